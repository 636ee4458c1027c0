import sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev, with_index_batch=False)
print(json.dumps(bench.index_build_from_strings_leg(dev, inp["model"], 3450000)))
