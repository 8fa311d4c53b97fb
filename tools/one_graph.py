"""one 256-atom graph per training step (bench.py: one_graph_leg) as a stand-alone run, for traces: python tools/one_graph.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space

dev = torch.device("cuda", 0)
hp = declare_gnn_space(HyperParameters(**bench.ARCH))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print(json.dumps(bench.one_graph_leg(dev, hp, steps=steps)))
