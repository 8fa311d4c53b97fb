"""the two small-call legs of bench.py alone (one graph per training step; one 7lgi frame per call)"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
dev = torch.device("cuda", 0)
hp = declare_gnn_space(HyperParameters(**bench.ARCH))
r = bench.one_graph_leg(dev, hp)
print(json.dumps({k: r[k] for k in ("ms_per_step", "ms_per_step_hipevent_median", "eager")}))
w = bench.whole_protein_leg(dev)
print(json.dumps({k: (v["ms_per_frame"] if isinstance(v, dict) and "ms_per_frame" in v else None) for k, v in w.items() if "frames_per_call" in k}))
