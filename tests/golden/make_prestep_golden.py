"""Golden vectors of the providers' host pre-step, produced by the REFERENCE's own functions.

    python tests/golden/make_prestep_golden.py          (build container only: needs /root/reference)

util.py imports tensorflow and modelnet_provider.py imports h5py at module level; neither exists in this image, so
the modules cannot be imported.  The four functions wanted here use numpy only, so their source text is cut out of
the reference files with `ast` (by function name), compiled and executed UNMODIFIED in a namespace that holds just
numpy.  Nothing of that text is stored: the fixtures hold inputs (seeded), the random draws the functions made
(re-derived from the same seed) and the functions' outputs.
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from pointwise_amd import synth  # noqa: E402


def reference_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
    missing = [n for n in names if n not in ns]
    if missing:
        raise SystemExit("reference functions not found: %s" % missing)
    return ns


def main():
    prov = reference_functions(os.path.join(REF, "modelnet_provider.py"),
                               ["rotate_point_cloud", "rotate_point_cloud_by_angle", "jitter_point_cloud"])
    util = reference_functions(os.path.join(REF, "util.py"), ["sort_point_cloud_xyz", "sort_point_cloud_xyz2"])
    # --- augmentation (modelnet_provider.py:196-198): rotate, then jitter, with a seeded global generator
    B, N = 4, 300
    P = synth.modelnet_like(B, N, seed=501, jitter=False)
    np.random.seed(777)
    rotated = prov["rotate_point_cloud"](P)
    jittered = prov["jitter_point_cloud"](rotated)
    np.random.seed(777)                       # the same draws, in the order the functions made them
    angles = np.array([np.random.uniform() * 2 * np.pi for _ in range(B)])
    noise = np.random.randn(B, N, 3)
    fixed = prov["rotate_point_cloud_by_angle"](P, 0.7)
    np.savez_compressed(os.path.join(HERE, "prestep_augment.npz"), points=P, angles=angles, noise=noise,
                        rotated=rotated, jittered=jittered, fed=jittered.astype(np.float32), fixed_angle=np.float64(0.7),
                        rotated_fixed=fixed)
    # --- sorting: generic cloud, a lattice cloud (many ties in x and in (x, y)), 9-channel S3DIS-like rows + labels
    cases = {}
    G = synth.uniform_cube(3, 257, 502)
    cases["generic"] = (G, None)
    L = synth.lattice(2, 400, 503, voxel=0.1, span=3, div=1)
    L = np.unique(L.reshape(-1, 3), axis=0)[:300][np.random.default_rng(5).permutation(300)][None].repeat(2, 0)
    L[1] = L[1][::-1]
    cases["lattice_unique"] = (np.ascontiguousarray(L), None)
    R = synth.room_like(2, 512, 504)
    F = synth.features(2, 512, 9, 505, points=R)
    lab = np.random.default_rng(6).integers(0, 13, size=(2, 512)).astype(np.uint8)
    cases["room9_labels"] = (F, lab)
    out = {}
    for name, (data, attr) in cases.items():
        out[name + "_in"] = data
        if attr is None:
            out[name + "_sorted"] = util["sort_point_cloud_xyz"](data)
        else:
            s, a = util["sort_point_cloud_xyz2"](data, attr)
            out[name + "_attr"] = attr
            out[name + "_sorted"] = s
            out[name + "_attr_sorted"] = a
    np.savez_compressed(os.path.join(HERE, "prestep_sort.npz"), **out)
    print("wrote prestep_augment.npz, prestep_sort.npz")


if __name__ == "__main__":
    main()
