#!/usr/bin/env python3
"""Root-level entry with the reference's file name: `python generate_data.py ...` (see distdiff_amd/generate_data.py)."""
import sys

from distdiff_amd.generate_data import main

if __name__ == "__main__":
    sys.exit(main())
