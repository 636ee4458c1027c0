#!/usr/bin/env python3
"""What does the encoder's 256-byte status block hold after HIP-graph replays?  The forward clears it with
hipMemsetAsync, then prep_len_kernel ORs status bits into word 0 and tt_absmax atomicMax-es max|W_ih| into words 40..;
tests/test_encoder_gpu.py::test_encoder_forward_replays_from_a_hip_graph failed once K1's scale came from those words."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests" / "golden"))
import torch
import synth
import twotowermlretrieval_amd as tt

V, E, H = 500, 52, 64
torch.manual_seed(0)
m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True},
                     synth.make_table(9, V, E)).cuda().eval()
enc = m.query_encoder
enc.check_inputs = False
ids = [torch.from_numpy(synth.make_ids(70 + s, 6, 11, V)).cuda() for s in range(3)]
B, T = ids[0].shape
al = lambda n: (n + 255) // 256 * 256
flag_off = al(4 * B) + al(4 * (B + 1)) + al(4 * B) + al(4 * B * T)


def words(ws):
    w = ws[flag_off:flag_off + 256].view(torch.int32).cpu().tolist()
    return {i: hex(v & 0xffffffff) for i, v in enumerate(w) if v}


with torch.no_grad():
    out, ws, st = enc._run_forward(ids[0], train=False)
    torch.cuda.synchronize()
    print("eager      :", words(ws), " max|W_ih l0| bits:", hex(enc.rnn.weight_ih_l0.abs().max().view(torch.int32).item()))
    want = [enc(x).clone() for x in ids]
    static = ids[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout, gws, gst = enc._run_forward(static, train=False)
    for rep in range(3):
        static.copy_(ids[rep])
        g.replay()
        torch.cuda.synchronize()
        print(f"replay {rep}   :", words(gws), " equal to eager:", torch.equal(gout, want[rep]),
              " max abs diff:", float((gout - want[rep]).abs().max()))
