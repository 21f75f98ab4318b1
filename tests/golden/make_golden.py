"""Generate the golden fixtures by running the REAL reference in the build container.

Run once, here (``python tests/golden/make_golden.py``): it imports the
reference operator from ``/root/reference/example`` (read-only, never copied,
never shipped), feeds it the seeded inputs of ``cases.py`` and stores inputs'
checksums, the replayed small arrays and the reference's intermediates/outputs
as ``tests/golden/<case>.npz``.  The GPU box has no ``/root/reference``; tests
only read the committed ``.npz`` files.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF = "/root/reference/example"


def import_reference():
    # transformer.py imports torch_geometric.nn.MLP at module level (not installed): stub it.
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")

    class MLP(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tgnn.MLP = MLP
    tg.nn = tgnn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgnn
    sys.path.insert(0, REF)
    import hept  # type: ignore
    import hept_utils  # type: ignore
    import transformer  # type: ignore

    return hept, hept_utils, transformer


def reference_forward(hept, hept_utils, inp, block_size, n_hashes):
    """Run the reference module and capture every intermediate by replaying its stages."""
    H, D, K = cases.NUM_HEADS, cases.H_DIM, cases.W_PER_DIST
    C = inp["coords"].shape[1]
    attn = hept.HEPTAttention(D + C, h_dim=D, num_heads=H, block_size=block_size, n_hashes=n_hashes, num_w_per_dist=K)
    with torch.no_grad():
        attn.e2lsh.alpha.copy_(inp["alpha"])
        attn.out_linear.weight.copy_(inp["out_weight"])
        attn.out_linear.bias.copy_(inp["out_bias"])
    w_rpe = torch.nn.Linear(K * (C - 1), H * D)
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    q, k, v, coords, codes = inp["q"], inp["k"], inp["v"], inp["coords"], inp["combined_shifts"]
    with torch.no_grad():
        out = attn(q, k, v, pe=coords, w_rpe=w_rpe, coords=coords, combined_shifts=codes)
        # stage replay (same calls, same order as the module's forward)
        from einops import rearrange

        qh, kh, vh = (t.view(-1, H, D) for t in (q, k, v))
        w = rearrange(w_rpe.weight, "(h d) (r k) -> h d r k", h=H, d=D, k=K)
        q_hat, k_hat = hept.prep_qk(qh, kh, w, coords)
        q_hat = rearrange(q_hat, "n h d -> h n d")
        k_hat = rearrange(k_hat, "n h d -> h n d")
        vh = rearrange(vh, "n h d -> h n d")
        q_hashed, k_hashed, hash_shift = hept_utils.lsh_mapping(attn.e2lsh, q_hat, k_hat)
        cs = codes * hash_shift
        q_keys, k_keys = q_hashed + cs, k_hashed + cs
        q_pos, k_pos = q_keys.argsort(dim=-1), k_keys.argsort(dim=-1)
        s_q = hept_utils.sort_to_buckets(q_hat, q_pos, block_size)
        s_k = hept_utils.sort_to_buckets(k_hat, k_pos, block_size)
        s_v = hept_utils.sort_to_buckets(vh, k_pos, block_size)
        denom, so = hept.qkv_res(s_q, s_k, s_v)
        rev = hept_utils.invert_permutation(q_pos)
        o = hept_utils.unsort_from_buckets(so, rev)
        logits = hept_utils.unsort_from_buckets(denom, rev)
        per_head = o.sum(dim=0) / logits.sum(dim=0)
        out2 = attn.out_linear(rearrange(per_head, "h n d -> n (h d)"))
    assert torch.equal(out, out2), "stage replay disagrees with the module forward"
    qw = w.sum(dim=1).clamp(max=50).exp().sum(dim=-1)
    sqrt_w = torch.sqrt(2 * torch.cat([qw[:, :1], qw], dim=-1))
    return dict(
        out=out, per_head=per_head, numer=o, denom=logits.squeeze(-1), q_positions=q_pos, k_positions=k_pos,
        q_hashed=q_hashed, k_hashed=k_hashed, hash_span=hash_shift, q_keys=q_keys, k_keys=k_keys, sqrt_w=sqrt_w,
        q_hat=q_hat, k_hat=k_hat,
    )


def reference_forward_fp64(hept, hept_utils, inp, block_size, q_pos, k_pos):
    """The reference's arithmetic evaluated in float64 on the SAME blocks: its own stage functions (dtype-agnostic torch
    code) on double inputs, with the permutations of the fp32 run injected -- what the operator's output would be without
    fp32 rounding.  Used where the fp32 reference itself is dominated by rounding (G7)."""
    from einops import rearrange

    H, D, K = cases.NUM_HEADS, cases.H_DIM, cases.W_PER_DIST
    q, k, v, coords = (inp[x].double() for x in ("q", "k", "v", "coords"))
    with torch.no_grad():
        qh, kh, vh = (t.view(-1, H, D) for t in (q, k, v))
        w = rearrange(inp["w_rpe_weight"].double(), "(h d) (r k) -> h d r k", h=H, d=D, k=K)
        q_hat, k_hat = hept.prep_qk(qh, kh, w, coords)
        q_hat = rearrange(q_hat, "n h d -> h n d")
        k_hat = rearrange(k_hat, "n h d -> h n d")
        vh = rearrange(vh, "n h d -> h n d")
        s_q = hept_utils.sort_to_buckets(q_hat, q_pos, block_size)
        s_k = hept_utils.sort_to_buckets(k_hat, k_pos, block_size)
        s_v = hept_utils.sort_to_buckets(vh, k_pos, block_size)
        denom, so = hept.qkv_res(s_q, s_k, s_v)
        rev = hept_utils.invert_permutation(q_pos)
        o = hept_utils.unsort_from_buckets(so, rev)
        logits = hept_utils.unsort_from_buckets(denom, rev)
        per_head = o.sum(dim=0) / logits.sum(dim=0)
        out = torch.nn.functional.linear(rearrange(per_head, "h n d -> n (h d)"), inp["out_weight"].double(),
                                         inp["out_bias"].double())
    assert out.dtype == torch.float64
    return out


def reference_gradients(hept, inp, block_size, n_hashes, seed=11):
    """Gradients of the REAL reference module (plain autograd) for a seeded upstream gradient."""
    H, D, K = cases.NUM_HEADS, cases.H_DIM, cases.W_PER_DIST
    C = inp["coords"].shape[1]
    attn = hept.HEPTAttention(D + C, h_dim=D, num_heads=H, block_size=block_size, n_hashes=n_hashes, num_w_per_dist=K)
    with torch.no_grad():
        attn.e2lsh.alpha.copy_(inp["alpha"])
        attn.out_linear.weight.copy_(inp["out_weight"])
        attn.out_linear.bias.copy_(inp["out_bias"])
    w_rpe = torch.nn.Linear(K * (C - 1), H * D)
    with torch.no_grad():
        w_rpe.weight.copy_(inp["w_rpe_weight"])
    q, k, v = (inp[x].clone().requires_grad_(True) for x in ("q", "k", "v"))
    out = attn(q, k, v, pe=inp["coords"], w_rpe=w_rpe, coords=inp["coords"], combined_shifts=inp["combined_shifts"])
    g_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed))
    out.backward(g_out)
    return dict(dq=q.grad, dk=k.grad, dv=v.grad, dw_rpe=w_rpe.weight.grad, dout_w=attn.out_linear.weight.grad,
                dout_b=attn.out_linear.bias.grad)


def key_monotonicity_digest(keys, pos):
    """Checks the reference permutation sorts its keys; returns per-(t,h) sums of sorted keys (float64)."""
    sk = torch.gather(keys, -1, pos)
    assert bool((sk[..., 1:] >= sk[..., :-1]).all())
    return sk.double().sum(-1).numpy()


def main():
    hept, hept_utils, transformer = import_reference()
    ckpt = torch.load("/root/reference/example/ckpt/tracking-60k-model.pt", map_location="cpu", weights_only=False)
    only = set(sys.argv[1:])   # python make_golden.py [case ...]: regenerate the named cases only
    for name, cfg in cases.CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(cfg["seed"])
        stored = {}
        T, B = cfg["n_hashes"], cfg["block_size"]
        if cfg.get("ckpt_weights"):
            stored["regions"] = ckpt["regions"].numpy()
            stored["w_rpe_weight"] = ckpt["attns.0.w_rpe.weight"].numpy()
            stored["alpha"] = ckpt["attns.0.attn.e2lsh.alpha"].numpy()
            stored["out_weight"] = ckpt["attns.0.attn.out_linear.weight"].numpy()
            stored["out_bias"] = ckpt["attns.0.attn.out_linear.bias"].numpy()
            for kk in ("w_q", "w_k", "w_v"):
                stored[kk] = ckpt[f"attns.0.{kk}.weight"].numpy()
            stored["norm1_weight"] = ckpt["attns.0.norm1.weight"].numpy()
            stored["norm1_bias"] = ckpt["attns.0.norm1.bias"].numpy()
        else:
            stored["regions"] = hept_utils.get_regions(cfg["num_regions"], T, cases.NUM_HEADS).numpy()
        # first pass: raw coords/batch of this case (our own padding, discarded)
        tmp = cases.build_inputs(name, stored)
        helper = {"block_size": B, "num_heads": cases.NUM_HEADS, "regions": torch.from_numpy(stored["regions"]).float()}
        n_raw = tmp["n_raw"]
        ref_pad, ref_kw, ref_unpad = transformer.prepare_input(
            torch.arange(n_raw), tmp["coords_raw"], tmp["batch"], helper
        )
        stored["pad_seq"] = ref_pad.numpy().astype(np.int32)
        inp = cases.build_inputs(name, stored)
        if not cfg.get("random_codes"):
            # our prepare_input mirror must reproduce the reference's AND codes; the only
            # admissible differences are exact ties in a coordinate straddling a region
            # boundary (the reference ranks with an unstable argsort) -> stored as patches
            bad = (inp["combined_shifts"] != ref_kw["combined_shifts"]).nonzero()
            assert len(bad) <= 8, (name, len(bad))
            stored["code_patch_idx"] = bad.numpy().astype(np.int32)
            stored["code_patch_val"] = ref_kw["combined_shifts"][tuple(bad.T)].numpy().astype(np.int64)
            inp = cases.build_inputs(name, stored)
            assert torch.equal(inp["combined_shifts"], ref_kw["combined_shifts"]), name
            print(f"  {name}: {len(bad)} tie-induced code patches")
        assert torch.equal(inp["unpad_seq"], ref_unpad), name
        assert torch.equal(inp["coords"], ref_kw["coords"]), name

        ref = reference_forward(hept, hept_utils, inp, B, T)
        fx = dict(stored)
        fx["input_checksums"] = cases.input_checksums(inp)
        fx["ref_codes_sum"] = np.asarray(ref_kw["combined_shifts"].double().sum().item())
        n = inp["q"].shape[0]
        fx["sorted_key_sums_q"] = key_monotonicity_digest(ref["q_keys"], ref["q_positions"])
        fx["sorted_key_sums_k"] = key_monotonicity_digest(ref["k_keys"], ref["k_positions"])
        fx["sqrt_w"] = ref["sqrt_w"].detach().numpy()
        fx["hash_span"] = ref["hash_span"].numpy()
        if cfg.get("sample_rows"):
            g = torch.Generator().manual_seed(cfg["seed"])
            rows = torch.randperm(n, generator=g)[: cfg["sample_rows"]].sort().values
            fx["rows"] = rows.numpy().astype(np.int32)
            fx["out_rows"] = ref["out"][rows].numpy()
            fx["per_head_rows"] = ref["per_head"][:, rows].numpy()
            fx["q_hashed_rows"] = ref["q_hashed"][..., rows].numpy()
            fx["k_hashed_rows"] = ref["k_hashed"][..., rows].numpy()
            fx["out_abs_mean"] = np.asarray(ref["out"].abs().mean().item())
        else:
            idx_t = np.uint16 if n < 65536 else np.int32
            fx["out"] = ref["out"].numpy()
            fx["q_positions"] = ref["q_positions"].numpy().astype(idx_t)
            fx["k_positions"] = ref["k_positions"].numpy().astype(idx_t)
            g = torch.Generator().manual_seed(cfg["seed"])
            rows = torch.randperm(n, generator=g)[:256].sort().values
            fx["rows"] = rows.numpy().astype(np.int32)
            fx["q_hashed_rows"] = ref["q_hashed"][..., rows].numpy()
            fx["k_hashed_rows"] = ref["k_hashed"][..., rows].numpy()
            fx["denom_rows"] = ref["denom"][..., rows].numpy()
            fx["per_head_rows"] = ref["per_head"][:, rows].numpy()
            if cfg.get("no_grads"):   # (a finite-output case: no gradients stored)
                # round 5: where the fp32 reference is rounding noise, pin the claim on its float64 evaluation
                fx["out_fp64"] = reference_forward_fp64(hept, hept_utils, inp, B, ref["q_positions"], ref["k_positions"]).numpy()
                path = os.path.join(HERE, name + ".npz")
                np.savez_compressed(path, **fx)
                print(f"{name}: N={n} out|mean|={ref['out'].abs().mean():.4f} finite={bool(torch.isfinite(ref['out']).all())} "
                      f"denom[min,max]=({ref['denom'].min():.3e},{ref['denom'].max():.3e}) "
                      f"-> {os.path.getsize(path)/1e3:.0f} kB")
                continue
            # reference gradients (same permutations: the module is deterministic) for the upstream gradient
            # randn(seed 11) used by tests/test_gpu_backward.py
            gr = reference_gradients(hept, inp, B, T)
            fx["ref_grad_rows"] = rows.numpy().astype(np.int32)
            fx["ref_dq_rows"] = gr["dq"][rows].numpy()
            fx["ref_dk_rows"] = gr["dk"][rows].numpy()
            fx["ref_dv_rows"] = gr["dv"][rows].numpy()
            fx["ref_dw_rpe"] = gr["dw_rpe"].numpy()
            fx["ref_dout_w"] = gr["dout_w"].numpy()
            if n <= 1024:
                fx["per_head"] = ref["per_head"].numpy()
                fx["q_hashed"] = ref["q_hashed"].numpy()
                fx["k_hashed"] = ref["k_hashed"].numpy()
                fx["denom"] = ref["denom"].numpy()
                fx["numer"] = ref["numer"].numpy()
                fx["q_keys"] = ref["q_keys"].numpy()
                fx["k_keys"] = ref["k_keys"].numpy()
                fx["codes"] = inp["combined_shifts"].numpy().astype(np.int32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: N={n} out|mean|={ref['out'].abs().mean():.4f} "
              f"denom[min,max]=({ref['denom'].min():.3e},{ref['denom'].max():.3e}) "
              f"-> {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__":
    main()
