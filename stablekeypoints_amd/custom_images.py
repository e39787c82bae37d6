"""Folder-of-images dataset with the reference's item contract {'img': float[3,S,S] in [0,1]}
(datasets/custom_images.py:20-25)."""
from __future__ import annotations

import os

import numpy as np
import torch


class CustomDataset(torch.utils.data.Dataset):
    EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")

    def __init__(self, data_root, image_size=512):
        self.files = sorted(os.path.join(data_root, f) for f in os.listdir(data_root) if f.lower().endswith(self.EXT))
        if not self.files:
            raise FileNotFoundError(f"no images under {data_root}")
        self.size = image_size

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        from PIL import Image
        img = Image.open(self.files[i]).convert("RGB").resize((self.size, self.size), Image.BILINEAR)
        return {"img": torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()}
