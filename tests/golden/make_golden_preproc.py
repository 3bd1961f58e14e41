#!/usr/bin/env python3
"""tests/golden/preproc_bicubic.npz: Pillow (the library the reference's data preparation calls,
datasets/tool/prepare_data.py:24-47) run on random and smooth uint8 images.
    python tests/golden/make_golden_preproc.py"""
import json
import os

import numpy as np
import PIL
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))
rs = np.random.RandomState(123)
cases = [(8, 16), (8, 32), (8, 64), (8, 128), (16, 32), (16, 64), (16, 128), (32, 64), (32, 128), (64, 128),
         (128, 16), (128, 8), (128, 32), (100, 37), (16, 16)]
arrs, meta = {}, []
for i, (a, b) in enumerate(cases):
    if i % 2 == 0:
        img = rs.randint(0, 256, (a, a, 3)).astype(np.uint8)
    else:   # smooth image with saturated regions (exercises clip8 on overshoot)
        y, x = np.mgrid[0:a, 0:a] / max(a - 1, 1)
        img = np.stack([255 * (x > 0.5), 255 * y, 255 * (np.sin(6 * x + 3 * y) > 0)], -1).astype(np.uint8)
    out = np.asarray(Image.fromarray(img, "RGB").resize((b, b), Image.BICUBIC))
    arrs[f"in{i}"], arrs[f"out{i}"] = img, out
    meta.append([a, b])
# the reference's LR -> SR chain on one image: HR 128 -> LR 16 -> SR 128
hr = rs.randint(0, 256, (128, 128, 3)).astype(np.uint8)
lr = Image.fromarray(hr, "RGB").resize((16, 16), Image.BICUBIC)
sr = lr.resize((128, 128), Image.BICUBIC)
arrs.update(chain_hr=hr, chain_lr=np.asarray(lr), chain_sr=np.asarray(sr))
arrs["meta"] = np.array(json.dumps({"cases": meta, "pillow": PIL.__version__}))
np.savez_compressed(os.path.join(OUT, "preproc_bicubic.npz"), **arrs)
print("wrote preproc_bicubic.npz", os.path.getsize(os.path.join(OUT, "preproc_bicubic.npz")) // 1024, "KiB")
