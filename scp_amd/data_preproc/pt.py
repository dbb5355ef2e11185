"""Point-cloud readers / writers of the encode path (drop-in for the input half of data_preproc/pt.py).

Only the formats the encode CLI touches: KITTI `.bin` (pt.py:190), ascii `.ply` (pt.py:224, the reference parses it line
by line and skips every line that does not start with three floats) and the ascii PLY writer used for `_quant.ply`.
PSNR / chamfer tooling (pc_error subprocess, KD-tree) is out of scope.
"""
import os

import numpy as np


def loadbin(file):
    """KITTI: float32 [P,4] -> (xyz [P,3], reflectance [P,1])."""
    points = np.fromfile(file, dtype=np.float32).reshape(-1, 4)
    return points[:, 0:3], points[:, 3:4]


def loadply(filedir, color_format="geometry"):
    """ascii PLY: every line whose first three tokens parse as floats is a point (header lines fail the parse)."""
    coords = []
    with open(filedir) as f:
        for line in f:
            w = line.split(" ")
            try:
                coords.append((float(w[0]), float(w[1]), float(w[2])))
            except (ValueError, IndexError):
                continue
    return np.array(coords).astype("float32").reshape(-1, 3), None


def pcread(path, color_format="geometry"):
    if not os.path.exists(path):
        raise Exception("no such file:" + path)
    if path.endswith(".ply"):
        return loadply(path, color_format)
    if path.endswith(".bin"):
        return loadbin(path)
    raise ValueError("unsupported point cloud format: " + path)


def ptread(path):
    return pcread(path, "geometry")[0]


def write_ply_data(filename, points):
    """ascii PLY with x y z float columns (what test_gene.py writes as `<name>_quant.ply`)."""
    points = np.asarray(points)
    with open(filename, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
                % len(points))
        for p in points:
            f.write("%s %s %s\n" % (repr(float(p[0])), repr(float(p[1])), repr(float(p[2]))))
