"""Synthetic VITON-HD / DressCode directory trees for the dataset-preprocessing tests (test infrastructure).  Deterministic: the golden
fixtures (tests/golden/dataset_ref.safetensors, produced by the REAL reference dataset classes on these trees) and the tests regenerate
the same files.  Images are written losslessly (PNG bytes under the .jpg names the loaders expect: PIL sniffs the content), so decoding does
not depend on the libjpeg build."""
import json
import os

import numpy as np
from PIL import Image


def _smooth_rgb(rng, h, w):
    low = rng.integers(0, 255, size=(h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
    return Image.fromarray(low).resize((w, h), Image.BILINEAR)


def _person(rng, h, w, ids, miss_wrist=None):
    """label map with the layout of a standing person + 18 COCO key-points (x, y) in this image's pixel frame"""
    lab = np.zeros((h, w), np.uint8)
    cx = w // 2 + int(rng.integers(-w // 16, w // 16))
    Y = lambda f: int(f * h)
    X = lambda f: int(cx + f * w)
    lab[Y(.05):Y(.10), X(-.08):X(.08)] = ids["hair"]
    lab[Y(.10):Y(.20), X(-.07):X(.07)] = ids["face"]
    lab[Y(.20):Y(.24), X(-.04):X(.04)] = ids["neck"]
    lab[Y(.24):Y(.55), X(-.16):X(.16)] = ids["upper"]
    lab[Y(.26):Y(.60), X(-.26):X(-.17)] = ids["r_arm"]
    lab[Y(.26):Y(.60), X(.17):X(.26)] = ids["l_arm"]
    lab[Y(.55):Y(.85), X(-.15):X(.15)] = ids["bottom"]
    lab[Y(.85):Y(.95), X(-.14):X(-.03)] = ids["r_leg"]
    lab[Y(.85):Y(.95), X(.03):X(.14)] = ids["l_leg"]
    lab[Y(.95):Y(.99), X(-.15):X(-.03)] = ids["r_shoe"]
    lab[Y(.95):Y(.99), X(.03):X(.15)] = ids["l_shoe"]
    if "bag" in ids:
        lab[Y(.50):Y(.62), X(.27):X(.36)] = ids["bag"]
    jit = lambda: float(rng.uniform(-0.01, 0.01))
    kp = {0: (0.0, .15), 1: (0.0, .23), 2: (-.17, .27), 3: (-.22, .42), 4: (-.22, .58), 5: (.17, .27), 6: (.22, .42), 7: (.22, .58),
          8: (-.08, .56), 9: (-.08, .72), 10: (-.08, .93), 11: (.08, .56), 12: (.08, .72), 13: (.08, .93), 14: (-.03, .13), 15: (.03, .13),
          16: (-.07, .14), 17: (.07, .14)}
    pts = np.array([[cx + (kp[i][0] + jit()) * w, (kp[i][1] + jit()) * h] for i in range(18)])
    if miss_wrist == "right":
        pts[4] = 0.0
    elif miss_wrist == "right+elbow":
        pts[4] = 0.0; pts[3] = 0.0
    elif miss_wrist == "left":
        pts[7] = 0.0
    pts[17] = 0.0                      # an undetected joint (ear): all-zero heat-map channel
    return lab, pts


VITON_IDS = dict(hair=2, face=13, neck=10, upper=5, r_arm=15, l_arm=14, bottom=9, r_leg=17, l_leg=16, r_shoe=19, l_shoe=18)
DC_IDS = dict(hair=2, face=11, neck=11, upper=4, r_arm=15, l_arm=14, bottom=6, r_leg=13, l_leg=12, r_shoe=10, l_shoe=9, bag=16)


def make_vitonhd(root, n=3, seed=0):
    """{root}/test/{image, cloth, image-parse-v3, openpose_json} + test_pairs.txt; source frame 768 x 1024 scaled down 4x (192 x 256 files:
    the loaders resize anyway) with key-points given in the 768 x 1024 frame like OpenPose's output"""
    rng = np.random.default_rng(seed)
    h, w = 256, 192
    for d in ("image", "cloth", "image-parse-v3", "openpose_json"):
        os.makedirs(os.path.join(root, "test", d), exist_ok=True)
    names = ["%05d_00.jpg" % i for i in range(n)]
    miss = [None, "right", "left", "right+elbow"]
    for i, nm in enumerate(names):
        _smooth_rgb(rng, h, w).save(os.path.join(root, "test", "image", nm), format="PNG")
        _smooth_rgb(rng, h, w).save(os.path.join(root, "test", "cloth", nm), format="PNG")
        lab, pts = _person(rng, h, w, VITON_IDS, miss[i % 4])
        Image.fromarray(lab).save(os.path.join(root, "test", "image-parse-v3", nm.replace(".jpg", ".png")))
        body25 = np.zeros((25, 3))
        rows = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18]
        body25[rows, 0] = pts[:, 0] * 4.0
        body25[rows, 1] = pts[:, 1] * 4.0
        body25[rows, 2] = (pts.sum(1) > 0) * 0.9
        with open(os.path.join(root, "test", "openpose_json", nm.replace(".jpg", "_keypoints.json")), "w") as f:
            json.dump({"people": [{"pose_keypoints_2d": body25.reshape(-1).tolist()}]}, f)
    with open(os.path.join(root, "test_pairs.txt"), "w") as f:
        for i, nm in enumerate(names):
            f.write("%s %s\n" % (nm, names[(i + 1) % n]))
    return names


def make_dresscode(root, per_category=2, seed=1):
    """{root}/{category}/{images, masks, label_maps, keypoints} + test_pairs_{paired,unpaired}.txt; key-points in the 384 x 512 frame"""
    rng = np.random.default_rng(seed)
    h, w = 256, 192
    out = {}
    miss = [None, "right", "left"]
    k = 0
    for cat in ("dresses", "upper_body", "lower_body"):
        base = os.path.join(root, cat)
        for d in ("images", "masks", "label_maps", "keypoints"):
            os.makedirs(os.path.join(base, d), exist_ok=True)
        ims = ["%06d_0.jpg" % (100 * (k + 1) + i) for i in range(per_category)]
        cls = ["%06d_1.jpg" % (100 * (k + 1) + i) for i in range(per_category)]
        ids = dict(DC_IDS)
        if cat == "dresses":
            ids["upper"] = 7; ids["bottom"] = 7
        for i, (im, cl) in enumerate(zip(ims, cls)):
            _smooth_rgb(rng, h, w).save(os.path.join(base, "images", im), format="PNG")
            _smooth_rgb(rng, h, w).save(os.path.join(base, "images", cl), format="PNG")
            m = np.zeros((h, w), np.uint8); m[h // 6:5 * h // 6, w // 5:4 * w // 5] = 255
            Image.fromarray(m).save(os.path.join(base, "masks", cl.replace(".jpg", ".png")))
            lab, pts = _person(rng, h, w, ids, miss[(k + i) % 3])
            Image.fromarray(lab).save(os.path.join(base, "label_maps", im.replace("_0.jpg", "_4.png")))
            kp = [[float(x * 2.0), float(y * 2.0), 0.9 if x + y > 0 else 0.0, float(j)] for j, (x, y) in enumerate(pts)]
            with open(os.path.join(base, "keypoints", im.replace("_0.jpg", "_2.json")), "w") as f:
                json.dump({"keypoints": kp}, f)
        for order in ("paired", "unpaired"):
            with open(os.path.join(base, "test_pairs_%s.txt" % order), "w") as f:
                for i, im in enumerate(ims):
                    f.write("%s %s\n" % (im, cls[i if order == "paired" else (i + 1) % per_category]))
        out[cat] = (ims, cls)
        k += 1
    return out
