"""`RandomAffineWithInverse` -- same interface as the reference's invertable_transform.py:6-92.

Draw order (four `torch.rand(1).item()` per image: angle, scale, tx, ty) and the theta layout
[[s cos, s sin, tx], [-s sin, s cos, ty]] follow invertable_transform.py:22-57 so a seeded run
produces the same augmentations.  The warp itself (affine_grid + bilinear grid_sample, zeros,
align_corners=False) runs on whatever device the tensor lives on -- images are warped on the GPU
instead of the reference's CPU round trip.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class RandomAffineWithInverse:
    def __init__(self, degrees=0, scale=(1.0, 1.0), translate=(0.0, 0.0)):
        self.degrees = degrees
        self.scale = scale
        self.translate = translate
        self.last_params = {"theta": torch.eye(2, 3).unsqueeze(0)}
        self.last_theta_host = self.last_params["theta"]     # host copy of last_params["theta"] (None if unknown)

    def create_affine_matrix(self, angle, scale, translations_percent):
        a = math.radians(angle)
        theta = torch.tensor([[math.cos(a), math.sin(a), translations_percent[0]],
                              [-math.sin(a), math.cos(a), translations_percent[1]]], dtype=torch.float)
        theta[:, :2] = theta[:, :2] * scale
        return theta.unsqueeze(0)

    def sample_theta(self, n: int) -> torch.Tensor:
        out = []
        for _ in range(n):
            angle = torch.rand(1).item() * (2 * self.degrees) - self.degrees
            sc = torch.rand(1).item() * (self.scale[1] - self.scale[0]) + self.scale[0]
            tr = (torch.rand(1).item() * (2 * self.translate[0]) - self.translate[0],
                  torch.rand(1).item() * (2 * self.translate[1]) - self.translate[1])
            out.append(self.create_affine_matrix(angle, sc, tr))
        return torch.cat(out, dim=0)

    def __call__(self, img_tensor, theta=None):
        if theta is None:
            theta = self.sample_theta(img_tensor.shape[0])
        # thetas are drawn (or handed in) on the host: keep that copy so consumers that need the numbers on the host
        # (the loss kernel takes the inverse affine as launch arguments) never read them back from the device
        self.last_theta_host = theta.detach() if theta.device.type == "cpu" else None
        theta = theta.to(img_tensor.device)
        self.last_params = {"theta": theta}
        grid = F.affine_grid(theta, img_tensor.size(), align_corners=False)
        return F.grid_sample(img_tensor, grid, align_corners=False)

    @staticmethod
    def invert(theta: torch.Tensor) -> torch.Tensor:
        last = torch.tensor([[0.0, 0.0, 1.0]], dtype=theta.dtype, device=theta.device).expand(theta.shape[0], -1, -1)
        return torch.inverse(torch.cat([theta, last], dim=1))[:, :2, :]

    def inverse(self, img_tensor):
        theta_inv = self.invert(self.last_params["theta"].to(img_tensor.device))
        grid = F.affine_grid(theta_inv, img_tensor.size(), align_corners=False)
        return F.grid_sample(img_tensor, grid, align_corners=False)
