"""Tooling (build container only): pack the public ta01-ta80 / dmu16-dmu20
benchmark instances into ``jssenv_amd/data/instances.npz``.

Input: the Taillard-format text files the reference ships as data
(JSSEnv/envs/instances/*, format described at jss_env.py:72-88).  Output: one
flat int32 array of ``machine << 16 | duration`` plus name/shape/offset
tables.  Parsed with this repo's own parser.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from jssenv_amd import instances as I  # noqa: E402
import refload  # noqa: E402


def main():
    src = os.path.dirname(refload.reference_instance_path("ta01"))
    names = sorted(os.listdir(src), key=lambda n: (n[:2] != "ta", n))
    ops, shapes, offsets = [], [], []
    off = 0
    for n in names:
        inst = I.load_instance_file(os.path.join(src, n))
        flat = inst.packed().reshape(-1)
        ops.append(flat)
        shapes.append((inst.jobs, inst.machines))
        offsets.append(off)
        off += flat.size
    out = os.path.join(os.path.dirname(HERE), "jssenv_amd", "data", "instances.npz")
    np.savez_compressed(out, names=np.array(names), shape=np.array(shapes, dtype=np.int32),
                        offset=np.array(offsets, dtype=np.int64), ops=np.concatenate(ops).astype(np.int32))
    print(f"{len(names)} instances, {off} ops -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
