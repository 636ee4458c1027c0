#!/usr/bin/env python3
"""Do host-to-device copies (shader kernels on this platform) stop serialising with the encoder when copy and compute streams
get disjoint CU masks (hipExtStreamCreateWithCUMask)?  16 batches of 16 384 passages, document tower:
  a   ids resident, default stream          a_m  ids resident, compute stream masked to all CUs but K
  b   pinned ids copied on a side stream    b_m  copies on a stream masked to the K CUs, compute on the masked compute stream"""
import sys, json, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
import bench

dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev, with_index_batch=False)
model = inp["model"].eval()
V, rs = bench.ENC_V, np.random.RandomState(3)
n_docs, bs = 262_144, 16384
host = []
for i in range(0, n_docs, bs):
    ids = bench.make_ids_bulk(rs, bs, 70, 10, 250, V)
    ids = ids[0] if isinstance(ids, tuple) else ids
    host.append(ids.pin_memory())
resident = [h.to(dev) for h in host]
hip = C.CDLL("libamdhip64.so")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
words = (cus + 31) // 32


def masked_stream(bits):
    m = (C.c_uint32 * words)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * w + 32) for w in range(words)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


def run(compute, copy):
    def f():
        with torch.no_grad(), torch.cuda.stream(compute):
            for k in range(len(host)):
                if copy is None:
                    ids = resident[k]
                else:
                    with torch.cuda.stream(copy):
                        ids = host[k].to(dev, non_blocking=True)
                    compute.wait_stream(copy)
                    ids.record_stream(compute)
                model.encode_document(ids)
    return f


default = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
print(json.dumps({"what": "a", "ms": round(timed(run(default, None)) * 1e3, 2)}), flush=True)
print(json.dumps({"what": "b", "ms": round(timed(run(default, side)) * 1e3, 2)}), flush=True)
for name, copy_bits in (("first K CUs", list(range(K))), ("one CU in every 32", [32 * w + j for w in range(words) for j in range(max(1, K // words))])):
    comp_bits = [b for b in range(cus) if b not in set(copy_bits)]
    cs, ks = masked_stream(comp_bits), masked_stream(copy_bits)
    print(json.dumps({"what": "a_m", "mask": name, "copy_cus": len(copy_bits), "ms": round(timed(run(cs, None)) * 1e3, 2)}), flush=True)
    print(json.dumps({"what": "b_m", "mask": name, "copy_cus": len(copy_bits), "ms": round(timed(run(cs, ks)) * 1e3, 2)}), flush=True)
print(json.dumps({"what": "a again", "ms": round(timed(run(default, None)) * 1e3, 2)}), flush=True)
