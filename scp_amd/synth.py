"""Synthetic LiDAR frames (test-input spec of SURVEY.md Appendix G / §8d).

No dataset is available offline, so every parity test, fixture and bench line in this
repo uses this generator: a 64-beam x 1875-azimuth HDL-64-like sweep over a flat ground
plane plus uniformly distributed obstacles.  P = 120 000 points per frame.
"""
import numpy as np


def synth_frame(seed=0, beams=64, az=1875):
    """xyz float32 [beams*az, 3]; RNG numpy PCG64 (default_rng)."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(2.0, -24.8, beams))
    phi = np.linspace(0, 2 * np.pi, az, endpoint=False)
    E, P = np.meshgrid(el, phi, indexing="ij")
    r_ground = np.where(E < 0, 1.73 / np.maximum(np.sin(-E), 1e-6), 120.0)
    r_obj = rng.uniform(5, 80, size=E.shape)
    r = np.minimum(r_ground, r_obj) + rng.normal(0, 0.02, size=E.shape)
    r = np.clip(r, 2.0, 120.0)
    x = r * np.cos(E) * np.cos(P)
    y = r * np.cos(E) * np.sin(P)
    z = r * np.sin(E)
    return np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)


def write_kitti_bin(path, xyz):
    """KITTI .bin layout: float32 [P,4] (x, y, z, reflectance=0)."""
    p = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1).astype(np.float32)
    p.tofile(path)


def ford_like(xyz):
    """Ford-like frame: same cloud in integer millimetres (float32 values)."""
    return np.round(xyz.astype(np.float64) * 1000).astype(np.float32)
