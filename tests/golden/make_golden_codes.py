"""Per-row digests of the REFERENCE's ``combined_shifts`` (example/transformer.py:35-63), for an elementwise check at test time.

``make_golden.py`` asserted elementwise equality of the host mirror's AND codes with the reference's at fixture time
and kept only a checksum; this script (run here, in the build container, where /root/reference exists) runs the
reference's own ``prepare_input`` on every case's raw inputs again and stores, per (table, head) row, a 64-bit digest
of the row's int64 codes (first 8 bytes of SHA-256 over the little-endian bytes) plus the row's sum and maximum in
``tests/golden/ref_codes_digest.npz``.  ``tests/test_prep_host.py`` recomputes the same digests from the codes the
tests feed to the operator: equal digests = equal rows, element for element.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from make_golden import import_reference  # noqa: E402


def row_digests(codes: torch.Tensor) -> np.ndarray:
    """(T, H, N) int64 -> (T, H) uint64: first 8 bytes of SHA-256 of each row's little-endian int64 bytes."""
    arr = np.ascontiguousarray(codes.numpy().astype("<i8"))
    t, h, _ = arr.shape
    out = np.empty((t, h), dtype=np.uint64)
    for i in range(t):
        for j in range(h):
            out[i, j] = int.from_bytes(hashlib.sha256(arr[i, j].tobytes()).digest()[:8], "little")
    return out


def main():
    _, _, transformer = import_reference()
    stored_all = {}
    for name, cfg in cases.CASES.items():
        if cfg.get("random_codes"):
            continue
        inp, fx = cases.load_case(name)
        helper = {"block_size": cfg["block_size"], "num_heads": cases.NUM_HEADS, "regions": inp["regions"]}
        _, ref_kw, _ = transformer.prepare_input(torch.arange(inp["n_raw"]), inp["coords_raw"], inp["batch"], helper)
        ref_codes = ref_kw["combined_shifts"]
        assert float(ref_codes.double().sum()) == float(fx["ref_codes_sum"]), name   # same run as make_golden.py's
        stored_all[name + "/digest"] = row_digests(ref_codes)
        stored_all[name + "/row_sum"] = ref_codes.sum(-1).numpy().astype(np.int64)
        stored_all[name + "/row_max"] = ref_codes.amax(-1).numpy().astype(np.int64)
        same = torch.equal(ref_codes, inp["combined_shifts"])
        print(f"{name}: reference codes {tuple(ref_codes.shape)}, equal to the replayed case inputs: {same}")
        assert same, name
    np.savez_compressed(os.path.join(HERE, "ref_codes_digest.npz"), **stored_all)


if __name__ == "__main__":
    main()
