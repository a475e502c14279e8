"""A rank program for tests/test_parallel_cpu.py::test_spawn_ranks_*: a gloo group on the CPU doing all-reduce "steps" the way
bench.py's ranks do (RANK / WORLD_SIZE / MASTER_* from the launcher's environment), with one rank dying on purpose.

    dp_kill_worker.py MODE [DIE_RANK]
        ok         every rank runs 5 steps and exits 0; rank 0 prints one JSON line
        midstep    DIE_RANK exits with status 7 after step 2, its peers are then inside / about to enter a collective
        noshow     DIE_RANK exits with status 9 before the rendezvous: its peers wait in init_process_group
        hang       every rank sleeps forever after init (the launcher's overall timeout must end the job)
"""
import datetime
import json
import os
import sys
import time


def main():
    mode = sys.argv[1]
    die_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if mode == 'noshow' and rank == die_rank:
        os._exit(9)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    # a long group timeout: nothing but the launcher's supervision may end a stuck job inside the test's time budget
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    if mode == 'hang':
        time.sleep(3600)
    t = torch.ones(1024)
    for step in range(5):
        if mode == 'midstep' and rank == die_rank and step == 2:
            os._exit(7)
        try:
            dist.all_reduce(t)
        except Exception:
            # gloo notices a closed peer connection; RCCL would not (it waits for its watchdog): model the worse case -- the
            # survivor stays stuck until the launcher ends it
            time.sleep(3600)
        t /= world
    if rank == 0:
        print(json.dumps({'ok': True, 'value': float(t[0])}), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
