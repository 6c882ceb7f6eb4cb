"""Hierarchical prototype builder, mirroring dataloader.py:664-747: guide features of every train image
(224x224, ImageNet mean/std) -> L2-normalise -> class mean = global prototype; average-linkage agglomerative clustering
into K groups per class -> group means. The features come from the HIP ResNet-50 (engine.guide_encode); the clustering
(`average_linkage_labels`) restates what the reference's sklearn call computes -- UPGMA on euclidean distances, the tree cut and the
label numbering of sklearn.cluster._agglomerative._hc_cut -- so the box needs neither sklearn nor scipy (tests/test_cli.py checks the
labels against sklearn where it is installed)."""
import heapq

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _load_image(path, size):
    from PIL import Image
    im = Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR)     # transforms.Resize((224,224)) default bilinear
    x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
    m, s = torch.tensor(IMAGENET_MEAN)[:, None, None], torch.tensor(IMAGENET_STD)[:, None, None]
    return (x - m) / s


def average_linkage_labels(X, K):
    """Labels of AgglomerativeClustering(n_clusters=K, linkage='average').fit(X).labels_ (dataloader.py:704-717).

    Average linkage (UPGMA): repeatedly merge the two closest clusters, d(k, i u j) = (n_i d(k, i) + n_j d(k, j)) / (n_i + n_j); the
    merges come out in non-decreasing distance, i.e. in scipy's dendrogram order (node n + m is the m-th merge, smaller child first).
    The cut into K clusters and the numbering of the labels follow sklearn's _hc_cut: split the youngest node K - 1 times with the same
    heap operations, label the leaves of the i-th heap entry with i."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    if K > n:
        raise ValueError("cannot cluster %d samples into %d groups" % (n, K))
    sq = (X * X).sum(1)
    D = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0))
    np.fill_diagonal(D, np.inf)
    ids = np.arange(n)                      # dendrogram node held by each live row
    size = np.ones(n)
    alive = np.ones(n, dtype=bool)
    children = np.zeros((max(n - 1, 0), 2), dtype=np.int64)
    for m in range(n - 1):
        flat = int(np.argmin(D))
        i, j = divmod(flat, n)
        if i > j:
            i, j = j, i
        a, b = int(ids[i]), int(ids[j])
        children[m] = (a, b) if a < b else (b, a)
        d = (size[i] * D[i] + size[j] * D[j]) / (size[i] + size[j])
        D[i, :] = d
        D[:, i] = d
        D[i, i] = np.inf
        D[j, :] = np.inf
        D[:, j] = np.inf
        size[i] += size[j]
        alive[j] = False
        ids[i] = n + m
    labels = np.zeros(n, dtype=np.int64)
    if K <= 1 or n <= 1:
        return labels

    def leaves(node):
        out, stack = [], [node]
        while stack:
            v = stack.pop()
            if v < n:
                out.append(v)
            else:
                stack.extend(children[v - n])
        return out

    nodes = [-(int(children[-1].max()) + 1)]
    for _ in range(K - 1):
        c = children[-nodes[0] - n]
        heapq.heappush(nodes, -int(c[0]))
        heapq.heappushpop(nodes, -int(c[1]))
    for i, node in enumerate(nodes):
        labels[leaves(-node)] = i
    return labels


def prototypes_from_features(feats, targets, num_classes, K):
    """dataloader.py:699-731 on L2-normalised features [N,D]."""
    feats = np.asarray(feats, dtype=np.float32)
    targets = np.asarray(targets)
    glob, loc = [], []
    for c in range(num_classes):
        f = feats[targets == c]
        glob.append(f.mean(axis=0))
        labels = average_linkage_labels(f, K)
        loc.append(np.stack([f[labels == k].mean(axis=0) for k in range(K)]))
    return np.array(glob), np.array(loc)


def extract_prototypes_with_encoder(args, engine, ds, batch_size=64, rank=0, world=1):
    """dataloader.py:734-747.  With world > 1 every rank encodes a contiguous shard of the training images and the features are
    all-gathered (rank order = dataset order), instead of every process repeating the whole pass as the reference's fan-out does;
    the clustering is deterministic, so every rank then derives identical prototypes."""
    size = engine.cfg.guide.input_size
    feats = []
    B = engine.B
    mine = list(range(len(ds)))
    if world > 1:
        from .launcher import shard_range
        mine = [i for i in shard_range(len(ds), world, rank) if i < len(ds)]
    for i0 in range(0, len(mine), B):
        paths = [ds.image_paths[i] for i in mine[i0:i0 + B]]
        x = torch.stack([_load_image(p, size) for p in paths])
        n = x.shape[0]
        if n < B:
            x = torch.cat([x, x[-1:].expand(B - n, -1, -1, -1)])
        f = engine.guide_encode(x.to(engine.device))[:n].float()
        f = f / f.norm(dim=-1, keepdim=True)                                       # dataloader.py:677
        feats.append(f.cpu())
    D = engine.cfg.guide.feature_dim
    feats = torch.cat(feats) if feats else torch.zeros(0, D)
    if world > 1:
        from .launcher import all_gather_rows
        feats = all_gather_rows(feats, device=engine.device)
    feats = feats.numpy()
    g, l = prototypes_from_features(feats, ds.targets.numpy(), len(ds.class_names), args.K)
    return torch.from_numpy(g), torch.from_numpy(l)
