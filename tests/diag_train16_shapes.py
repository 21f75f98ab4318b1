"""Diagnostic (not a test): fp32 vs bf16 training tiles on synthetic clouds of several shapes, clustered (cluster_size 8)
and not: max error / tensor scale and mean error / mean magnitude per tensor.  python tests/diag_train16_shapes.py"""
import os, sys, torch
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (R, R+"/tests", R+"/tests/golden", R+"/oracle"): sys.path.insert(0,p)
from hept_amd.synthetic import make_inputs
from test_gpu_backward import _train_once
dev=torch.device("cuda",0)
for heads,d,c,cl in ((8,24,6,8),(4,16,4,8),(16,24,6,8),(5,20,5,8),(4,16,4,0),(8,24,6,0),(8,16,4,8),(4,24,6,8)):
    inp = make_inputs([700, 420], block_size=64, n_hashes=2, coords_dim=c, h_dim=d, num_heads=heads, seed=31, cluster_size=cl)
    inp["block_size"]=64
    ref=_train_once(inp,"fp32",dev); got=_train_once(inp,"bf16",dev)
    print(heads,d,c,cl," ".join("%s %.2e/%.2e"%(nm,float((a-b).abs().max()/b.abs().max()), float((a-b).abs().mean()/b.abs().mean())) for nm,a,b in zip(("out","dq","dk","dv","dwr","dWo"),got,ref)), flush=True)
