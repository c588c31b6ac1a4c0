#!/usr/bin/env python3
"""Where the PCIe-inclusive step goes (value_pcie_inclusive < value_pcie_serial on the final build).

rocprofv3 --kernel-trace --memory-copy-trace of bench.py dies in its own finaliser on this pool (SIGSEGV inside __cxa_finalize, no
output written: gpurun_out/pcie_trace.err of round 4), so the timeline is taken in-process: bench.PciePipeline.run(trace=True)
brackets every operation of the pipelined step -- the four transfers and the two compute calls -- with timing events on the
stream the operation is issued to.  Reported: each operation's duration ALONE (nothing else on the GPU) and INSIDE the pipeline,
the period of the compute stream (encode(k) start -> encode(k + 1) start), the time the compute stream sits idle between its
two calls, and which operation ends last before each compute call starts (what it waited for).

    python tools/pcie_timeline.py [--steps 8] [--batch 24] > gpurun_out/pcie_timeline.json
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    import torch
    import bench
    from llicti_amd.codec import HipCodec, mode_of_name
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, H, W = a.batch, 512, 768
    name = bench.default_container(H, W)
    mode = mode_of_name(name)
    torch.manual_seed(1337)
    codec = HipCodec(dev)
    codec.load_state_dict(LLICTI(default_config()).state_dict())
    rgb_h = bench.make_batch(B, H, W, 0)
    rgb = torch.from_numpy(rgb_h).to(dev)
    if mode & 0x10000:                      # encoder mode "auto": time the container it writes on this batch (the decoder needs the container's own mode)
        c0, _ = codec.encode(rgb, mode=mode)
        mode = sorted(set(codec.container_modes(c0)))[0]
        del c0
    stride = codec.max_container_bytes(H, W)
    cont = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    seg = torch.zeros((B, 49), dtype=torch.int32, device=dev)
    rec = torch.empty_like(rgb)
    rgb_pin = torch.from_numpy(rgb_h).pin_memory()
    cont_pin = torch.empty((B, stride), dtype=torch.uint8).pin_memory()
    seg_pin = torch.empty((B, 49), dtype=torch.int32).pin_memory()
    rec_pin = torch.empty((B, 3, H, W), dtype=torch.uint8).pin_memory()

    def ev_ms(fn, n=5):
        fn()
        torch.cuda.synchronize()
        out = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1))
        return round(statistics.median(out), 4)
    alone = {
        "encode": ev_ms(lambda: codec.encode(rgb, mode=mode, out=cont, seg_len=seg)),
        "decode": ev_ms(lambda: codec.decode(cont, seg, H, W, mode=mode, out=rec)),
        "h2d_rgb": ev_ms(lambda: rgb.copy_(rgb_pin, non_blocking=True)),
        "d2h_containers": ev_ms(lambda: (cont_pin.copy_(cont, non_blocking=True), seg_pin.copy_(seg, non_blocking=True))),
        "h2d_containers": ev_ms(lambda: (cont.copy_(cont_pin, non_blocking=True), seg.copy_(seg_pin, non_blocking=True))),
        "d2h_rgb": ev_ms(lambda: rec_pin.copy_(rec, non_blocking=True)),
    }
    nbytes = {"h2d_rgb": rgb.numel(), "d2h_containers": cont.numel() + seg.numel() * 4, "h2d_containers": cont.numel() + seg.numel() * 4, "d2h_rgb": rec.numel()}
    pipe = bench.PciePipeline(torch, codec, dev, mode, rgb, cont, seg, rec, rgb_pin, cont_pin, seg_pin, rec_pin)
    pipe.run(2)
    # wall clock of untraced runs (the bench's estimator), repeated
    est = []
    for _ in range(a.repeats):
        t = []
        for n in (3, 3 + a.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run(n)
            t.append(time.perf_counter() - t0)
        est.append((t[1] - t[0]) / a.steps * 1e3)
    # host-side cost of issuing one pipelined step (no GPU wait inside run() except its final synchronise)
    _, spans = pipe.run(3 + a.steps, trace=True)
    summary = bench.summarize_pcie_spans(spans)
    ops = {}
    for sp in spans:
        ops.setdefault(sp["op"], []).append(sp)
    inside = {k: round(statistics.median(s["end_ms"] - s["start_ms"] for s in v[2:]), 4) for k, v in ops.items()}
    enc = sorted(ops["encode"], key=lambda s: s["step"])
    dec = sorted(ops["decode"], key=lambda s: s["step"])
    period = [enc[i + 1]["start_ms"] - enc[i]["start_ms"] for i in range(2, len(enc) - 1)]
    # compute-stream idle time per step: gaps between the end of one compute call and the start of the next
    calls = sorted(enc + dec, key=lambda s: s["start_ms"])
    gaps = [round(calls[i + 1]["start_ms"] - calls[i]["end_ms"], 4) for i in range(4, len(calls) - 1)]
    # what each compute call waited for: the latest-ending transfer that ends within 0.2 ms before the call starts
    blame = {}
    for c in calls[4:]:
        cand = [s for s in spans if s["op"] not in ("encode", "decode") and c["start_ms"] - 0.2 <= s["end_ms"] <= c["start_ms"] + 0.05]
        key = f'{c["op"]} <- ' + (max(cand, key=lambda s: s["end_ms"])["op"] if cand else "previous compute call")
        blame[key] = blame.get(key, 0) + 1
    out = {
        "workload": f"{B}x{W}x{H} uniform-noise RGB, container {name}", "mpix_per_step": B * H * W / 1e6, "summary": summary,
        "alone_ms": alone, "inside_pipeline_ms": inside,
        "transfer_GBps_alone": {k: round(nbytes[k] / alone[k] / 1e6, 1) for k in nbytes},
        "transfer_GBps_inside": {k: round(nbytes[k] / inside[k] / 1e6, 1) for k in nbytes},
        "bytes": nbytes,
        "compute_period_ms": {"median": round(statistics.median(period), 4), "min": round(min(period), 4), "max": round(max(period), 4)},
        "compute_idle_gap_ms": {"sum_per_step_median": round(2 * statistics.median(gaps), 4), "max": max(gaps), "all": gaps},
        "compute_call_started_after": blame,
        "wall_estimator_ms_per_step": [round(e, 3) for e in sorted(est)],
        "resident_ms_per_step": round(alone["encode"] + alone["decode"], 4),
        "serial_ms_per_step": round(sum(alone.values()), 4),
        "mpix_s": {"resident": round(B * H * W / (alone["encode"] + alone["decode"]) / 1e3, 1),
                   "serial_sum_of_alone": round(B * H * W / sum(alone.values()) / 1e3, 1),
                   "pipelined_traced_period": round(B * H * W / statistics.median(period) / 1e3, 1),
                   "pipelined_wall_median": round(B * H * W / statistics.median(est) / 1e3, 1)},
        "spans_last_steps": [s for s in spans if s["step"] >= a.steps],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
