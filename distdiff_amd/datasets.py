"""Train-split listings of the datasets the expansion step reads: (image_paths, labels, class_names) exactly as the
reference's `StandardDataLoader.load_dataset()` produces them for `SDDataset` (dataloader.py:95-130, 750-755).

Only the listings BASELINE.json's configs need are built: caltech-101 (dataloader.py:272-315),
stanford_cars (dataloader.py:167-228) and a generic `<root>/<dataset>/train/<category>/` tree with the same rules.  The order
matters: the latent cache `image_latents.pt` is a positional list (dataloader.py:788-796) and the shard function cuts the
class-sorted list into contiguous ranges (generate_data.py:1001-1009).
"""
import os

DATASET_PATH = "{root}/{name}"      # dataloader.py:64  './data/{}'


def _class_dir_listing(train_path, skip=()):
    """`sorted(os.listdir(train))` categories, then each category's files in `os.listdir` order (the reference does not sort
    the files, dataloader.py:283-287)."""
    categories = sorted(os.listdir(train_path))
    categories = [c for c in categories if c not in skip and os.path.isdir(os.path.join(train_path, c))]
    paths, labels = [], []
    for i, c in enumerate(categories):
        files = [os.path.join(train_path, c, x) for x in os.listdir(os.path.join(train_path, c))]
        paths.extend(files)
        labels.extend([i] * len(files))
    return paths, labels, categories


def list_caltech101(root):
    """dataloader.py:272-315: `./data/caltech-101/train/<category>`; BACKGROUND_Google and Faces_easy dropped; 100 classes."""
    train = os.path.join(DATASET_PATH.format(root=root, name="caltech-101"), "train")
    paths, labels, cats = _class_dir_listing(train, skip=("BACKGROUND_Google", "Faces_easy"))
    assert len(cats) == 100, "caltech-101: expected 100 train categories under %s, found %d" % (train, len(cats))
    return paths, labels, cats


def list_stanford_cars(root):
    """dataloader.py:167-228: devkit/cars_train_annos.mat + devkit/cars_meta.mat; images under cars_train/; class names
    'Make Model Type Year' become 'Year Make Model Type'; classes ordered by label; 196 classes."""
    from scipy import io
    base = DATASET_PATH.format(root=root, name="stanford_cars")
    anno = io.loadmat(os.path.join(base, "devkit", "cars_train_annos.mat"))["annotations"][0]
    meta = io.loadmat(os.path.join(base, "devkit", "cars_meta.mat"))["class_names"][0]
    paths, labels, name_to_label = [], [], {}
    for a in anno:
        imname = a["fname"][0]
        label = int(a["class"][0, 0]) - 1
        names = str(meta[label][0]).split(" ")
        year = names.pop(-1)
        names.insert(0, year)
        classname = " ".join(names)
        name_to_label.setdefault(classname, label)
        paths.append(os.path.join(base, "cars_train", str(imname)))
        labels.append(label)
    class_names = [k for k, _ in sorted(name_to_label.items(), key=lambda kv: kv[1])]
    assert len(class_names) == 196, "stanford_cars: expected 196 classes, found %d" % len(class_names)
    return paths, labels, class_names


def list_generic(root, name):
    """`<root>/<dataset>/train/<category>/` as the reference's imagenette loader reads it (dataloader.py:317-331).  ImageNet subsets
    (BASELINE.json configs[4]: ImageNet-100) name their categories by WordNet id: an optional `<root>/<dataset>/classnames.txt`
    ("<directory name> <class name>" per line, the format of ImageNet's LOC_synset_mapping.txt) supplies the names the prompts use."""
    base = DATASET_PATH.format(root=root, name=name)
    train = os.path.join(base, "train")
    paths, labels, cats = _class_dir_listing(train if os.path.isdir(train) else base)
    mapping = os.path.join(base, "classnames.txt")
    if os.path.exists(mapping):
        names = {}
        for line in open(mapping):
            parts = line.strip().split(None, 1)
            if len(parts) == 2:
                names[parts[0]] = parts[1].split(",")[0].strip()
        cats = [names.get(c, c) for c in cats]
    return paths, labels, cats


def load_train_listing(name, root="data"):
    """-> (image_paths, labels, class_names) with `_` -> ' ' in the class names (dataloader.py:129)."""
    if name == "caltech-101":
        paths, labels, names = list_caltech101(root)
    elif name == "stanford_cars":
        paths, labels, names = list_stanford_cars(root)
    else:
        base = DATASET_PATH.format(root=root, name=name)
        if not os.path.isdir(base):
            raise SystemExit("dataset directory %s not found (listings built: caltech-101, stanford_cars, or a "
                             "<root>/<dataset>/train/<category>/ tree; or use --synthetic N)" % base)
        paths, labels, names = list_generic(root, name)
    if not paths:
        raise SystemExit("dataset %s: no training images found under %s" % (name, DATASET_PATH.format(root=root, name=name)))
    return paths, labels, [s.replace("_", " ") for s in names]
