"""On-disk formats of the reference that sit either side of the hot path — host-side only, byte-compatible
with what the reference writes and reads, so files can be exchanged with an unmodified checkout:

  * pre-computed CLIP cloth features: `data/clip_cloth_embeddings/{dataset}/{split}_last_hidden_state_features.pt` (one fp16 tensor
    [N, 257, 1280], torch.save) + `{split}_features_names.pkl` (pickled list of cloth file names) — written by
    src/utils/compute_cloth_clip_features.py:158-166, read by src/dataset/vitonhd.py:100-107 (and dresscode.py likewise);
  * released checkpoints: plain `state_dict` .pth files (hubconf.py:25-26,35-36,51-52; the warping file holds {'tps', 'refinement'},
    :60-62) -> the dict the native module constructors take;
  * generated images: `{save_dir}/{category}/{im_name}` as JPEG quality 95 or PNG (src/inference.py:314-324).
"""
import os
import pickle

import torch


def clip_feature_paths(root, dataset, split):
    d = os.path.join(root, "data", "clip_cloth_embeddings", dataset)
    return os.path.join(d, "%s_last_hidden_state_features.pt" % split), os.path.join(d, "%s_features_names.pkl" % split)


def save_clip_cloth_features(root, dataset, split, last_hidden_states, cloth_names):
    """last_hidden_states: [N, 257, 1280] (any float dtype / device; stored as CPU fp16 like compute_cloth_clip_features.py:158)"""
    if last_hidden_states.dim() != 3 or last_hidden_states.shape[0] != len(cloth_names):
        raise ValueError("features must be [N, tokens, hidden] with one row per cloth name")
    ft, nm = clip_feature_paths(root, dataset, split)
    os.makedirs(os.path.dirname(ft), exist_ok=True)
    torch.save(last_hidden_states.detach().to("cpu", torch.float16).contiguous(), ft)
    with open(nm, "wb") as f:
        pickle.dump(list(cloth_names), f)
    return ft, nm


def load_clip_cloth_features(root, dataset, split):
    """-> (features [N, tokens, hidden] CPU tensor, names list); per-item lookup as in vitonhd.py:150-152:
    `features[names.index(c_name)]`"""
    ft, nm = clip_feature_paths(root, dataset, split)
    feats = torch.load(ft, map_location="cpu").detach().requires_grad_(False)
    with open(nm, "rb") as f:
        names = pickle.load(f)
    if feats.shape[0] != len(names):
        raise ValueError("feature file and name file disagree (%d vs %d)" % (feats.shape[0], len(names)))
    return feats, names


def load_released_state_dict(path, key=None):
    """a released .pth checkpoint (hubconf.py) -> state_dict for the native module constructors; `key` selects a sub-dict
    ('tps' / 'refinement' of the warping checkpoint)"""
    sd = torch.load(path, map_location="cpu")
    if key is not None:
        sd = sd[key]
    if not isinstance(sd, dict) or not all(torch.is_tensor(v) for v in sd.values()):
        raise ValueError("%s does not hold a plain state_dict" % path)
    return sd


def save_generated_images(images, save_dir, categories, im_names, use_png=False):
    """images: list of PIL images (pipeline output_type='pil'); layout and encoder settings of inference.py:314-324"""
    out = []
    for img, cat, name in zip(images, categories, im_names):
        os.makedirs(os.path.join(save_dir, cat), exist_ok=True)
        if use_png:
            name = name.replace(".jpg", ".png")
            img.save(os.path.join(save_dir, cat, name))
        else:
            img.save(os.path.join(save_dir, cat, name), quality=95)
        out.append(os.path.join(save_dir, cat, name))
    return out
