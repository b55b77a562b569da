import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
for wl in sys.argv[1:]:
    for dt in ("bf16", "f32"):
        r = bench.slice_step(wl, dev, steps=100, warmup=10, dtype_name=dt)
        print(wl, dt, r["ms_per_step"], "ms/step", r["images_per_s"], "img/s", flush=True)
