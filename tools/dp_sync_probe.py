"""Which part of an overlapped all-reduce hand-over stalls the compute stream on MI355X (one rank, RCCL): a chain of kernels with
a hand-over every K kernels.  usage: python tools/dp_sync_probe.py"""
import os, time
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.randn(16 << 20, device=dev)
y = torch.empty_like(x)
g = torch.zeros(8 << 20, device=dev)        # 32 MiB bucket
comm = torch.cuda.Stream()
dist.all_reduce(g); torch.cuda.synchronize()


def handover(kind):
    main = torch.cuda.current_stream()
    if kind == 'events_only':
        e = torch.cuda.Event(); e.record(main); comm.wait_event(e)
        d = torch.cuda.Event(); d.record(comm)
        return d
    if kind == 'events+kernel':
        e = torch.cuda.Event(); e.record(main); comm.wait_event(e)
        with torch.cuda.stream(comm):
            g.add_(1.0)
        d = torch.cuda.Event(); d.record(comm)
        return d
    if kind == 'allreduce':
        e = torch.cuda.Event(); e.record(main); comm.wait_event(e)
        with torch.cuda.stream(comm):
            dist.all_reduce(g)
        d = torch.cuda.Event(); d.record(comm)
        return d
    if kind == 'allreduce_async':
        e = torch.cuda.Event(); e.record(main); comm.wait_event(e)
        with torch.cuda.stream(comm):
            w = dist.all_reduce(g, async_op=True)
        return w
    if kind == 'allreduce_main':               # on the compute stream itself
        dist.all_reduce(g)
        return None


def run(kind, n=400, every=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = []
    for i in range(n):
        torch.mul(x, 1.0001, out=y)
        if kind != 'none' and i % every == every - 1:
            pend.append(handover(kind))
    for d in pend:
        if isinstance(d, torch.cuda.Event):
            torch.cuda.current_stream().wait_event(d)
        elif d is not None:
            d.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n


for kind in ('none', 'events_only', 'events+kernel', 'allreduce', 'allreduce_async', 'allreduce_main', 'none', 'allreduce'):
    run(kind, 50)
    base = run(kind)
    print(f'{kind:16s} {base:7.2f} us per kernel (hand-over every 10 kernels)', flush=True)
dist.destroy_process_group()
