#!/usr/bin/env python3
"""Launch latency of a ONE-rank RCCL collective at the message sizes of the sharded sweep
(s x R fp64 partial MTTKRP rows: 16 KB at cfg2, 64 KB at cfg4), in-stream, back to back: the
fixed cost every mode update of a P-GPU run pays before any byte crosses xGMI. One rank is all a
one-GPU box offers; the P-rank latency adds the ring/tree hops on top (not measurable here).

    python tools/rccl_latency.py            (prints a small table; run on a GPU box)
"""
import os
import time

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print(f"{'collective':16s} {'bytes':>8s} {'us/call (stream, back to back)':>32s} {'us/call (host-synchronised)':>30s}")
    for nbytes in (16 << 10, 32 << 10, 64 << 10, 1 << 20):
        n = nbytes // 8
        x = torch.ones(n, dtype=torch.float64, device="cuda")
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        ops = {
            "all_reduce": lambda: dist.all_reduce(x),
            "reduce_scatter": lambda: dist.reduce_scatter_tensor(y, x),
            "all_gather": lambda: dist.all_gather_into_tensor(y, x),
        }
        for name, fn in ops.items():
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 500
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            stream_us = e0.elapsed_time(e1) * 1e3 / reps
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
                torch.cuda.synchronize()
            host_us = (time.perf_counter() - t0) * 1e6 / 100
            print(f"{name:16s} {nbytes:8d} {stream_us:32.2f} {host_us:30.2f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
