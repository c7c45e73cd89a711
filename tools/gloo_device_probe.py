#!/usr/bin/env python3
"""Two ranks on one GPU, gloo backend: what happens to an all-reduce of a device tensor?
Each case runs in its own pair of child processes so that an abort in one does not hide the next.
    python tools/gloo_device_probe.py            (parent: runs every case)
"""
import os
import subprocess
import sys

CASES = ["host", "device_sync", "device_async", "device_view_async", "big_sync", "two_async", "hook_then_bucket"]


def child(case, rank, port):
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    dev = "cpu" if case == "host" else "cuda:0"
    if case in ("big_sync", "two_async", "hook_then_bucket"):
        n = 8 << 20                                   # one 32 MiB bucket of the trainer
        flat = torch.full((2 * n + 1000,), float(rank + 1), device=dev)
        if case == "big_sync":
            dist.all_reduce(flat[:n], op=dist.ReduceOp.SUM)
        elif case == "two_async":
            works = [dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, async_op=True) for o in range(0, flat.numel(), n)]
            for w in works:
                w.wait()
        else:                                         # an early (hook) reduce left pending while compute goes on
            w0 = dist.all_reduce(flat[:n], op=dist.ReduceOp.SUM, async_op=True)
            a = torch.randn(4096, 4096, device=dev)
            for _ in range(10):
                a = a @ a * 1e-3
            works = [dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, async_op=True) for o in range(n, flat.numel(), n)]
            for w in works:
                w.wait()
            w0.wait()
        torch.cuda.synchronize()
        print(f"case {case} rank {rank}: ok, value {float(flat[0])}", flush=True)
        dist.destroy_process_group()
        return
    flat = torch.full((1 << 20,), float(rank + 1), device=dev)
    t = flat[128:128 + (1 << 18)] if case == "device_view_async" else flat
    if case.endswith("async"):
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        w.wait()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if dev != "cpu":
        torch.cuda.synchronize()
    print(f"case {case} rank {rank}: ok, value {float(t[0])}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 4:
        child(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    import torch
    print("gloo available:", torch.distributed.is_gloo_available(), "torch", torch.__version__, flush=True)
    for i, case in enumerate(CASES):
        procs = [subprocess.Popen([sys.executable, __file__, case, str(r), str(29600 + i)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, env=dict(os.environ, TORCH_SHOW_CPP_STACKTRACES="1"))
                 for r in range(2)]
        for r, p in enumerate(procs):
            try:
                out, _ = p.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
                out += b"\n[probe] TIMEOUT after 90 s\n"
            text = out.decode(errors="replace")
            keep = [ln for ln in text.splitlines() if "amdgpu.ids" not in ln]
            print(f"--- case {case} rank {r}: exit {p.returncode}\n" + "\n".join(keep[-25:]), flush=True)
