#!/usr/bin/env python3
"""Interleaved A/B of the train step's forward ordering (trainer._FWD_ORDER) on one box: eager step time per mode, three rounds;
the parameters after the steps must be the same bits in every mode (the recurrence kernels are bit-identical).
With a mode name as argument: six steps of that mode only (for tools/train_timeline.sh-style kernel traces)."""
import json, sys, time, copy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd import trainer


def main():
    dev = torch.device("cuda:0")
    inp = bench.make_encoder_inputs(dev, with_index_batch=False)
    m0 = inp["model"].train()
    q, p, n = (inp[k].to(dev) for k in "qpn")
    modes = sys.argv[1:] or ["query_first", "doc_first", "query_one_wg"]
    res, params = {k: [] for k in modes}, {}
    for rep in range(3 if len(modes) > 1 else 1):
        for mode in modes:
            trainer._FWD_ORDER = mode
            m = copy.deepcopy(m0)
            opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
            for _ in range(3):
                tt.train_step(m, opt, q, p, n, margin=0.5, defer_check=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tt.train_step(m, opt, q, p, n, margin=0.5, defer_check=True)
            opt.settle()
            torch.cuda.synchronize()
            res[mode].append(round((time.perf_counter() - t0) / 10 * 1e3, 4))
            params.setdefault(mode, opt.flat_params.detach().clone())
    trainer._FWD_ORDER = "query_first"
    same = all(torch.equal(params[modes[0]], params[k]) for k in modes)
    print(json.dumps({"eager_deferred_ms_per_step": res, "same_parameters_in_every_mode": same}), flush=True)


if __name__ == "__main__":
    main()
