"""distdiff_amd — MI355X-native guided-diffusion expansion engine (drop-in for DistDiff generate_data.py hot path)."""
