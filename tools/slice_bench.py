"""The layer-loop slice step (mga_yolo_amd/slice.py) alone: ms/step from graph replay; run under rocprofv3 --kernel-trace --stats for the
per-kernel split.    python tools/slice_bench.py [workload] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
print(bench.slice_step(wl, torch.device("cuda", 0), steps=steps))
