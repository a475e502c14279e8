"""How well do the two streams of the step fill the chip?  Reads a rocprofv3 --kernel-trace CSV of a bench.py run and, over the
steady-state window (the last `nsteps` steps, delimited by the k_adam launches of the generator), reports per step:
  wall time, time with 0 / 1 / >= 2 kernels in flight, and per kernel family the time it ran ALONE vs beside another kernel.

  cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o t --output-format csv -- python3 $R/bench.py --steps 12 --warmup 5 \
      --no-cpu-baseline --no-extra --events none
  python tools/timeline_overlap.py gpurun_out/tl [nsteps]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r'(k_[a-z0-9_]+)', name)
    return m.group(1) if m else name[:40]


def main():
    root = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
    assert files, 'no kernel trace under ' + root
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '?'))))
    rows.sort()
    # step boundaries: the generator's Adam launch = the LARGER of the two k_adam launches of a step (41.8 M vs 2.8 M parameters)
    adams = [(s, e) for s, e, n, q in rows if 'k_adam' in n]
    big = sorted(adams, key=lambda t: t[1] - t[0])[len(adams) // 2:]
    big.sort()
    marks = [s for s, e in big][-(nsteps + 1):]
    t0, t1 = marks[0], marks[-1]
    win = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in rows if e > t0 and s < t1]
    ev = []
    for i, (s, e, n, q) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active = set()
    last = t0
    depth_time = defaultdict(float)
    solo = defaultdict(float)
    shared = defaultdict(float)
    queues = defaultdict(float)
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            depth_time[min(len(active), 2)] += dt
            for j in active:
                (solo if len(active) == 1 else shared)[family(win[j][2])] += dt
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    for s, e, n, q in win:
        queues[q] += e - s
    ms = lambda x: x / nsteps / 1e6
    print(f'{nsteps} steps, {ms(t1 - t0):.3f} ms per step; kernels in flight: none {ms(depth_time[0]):.3f} ms, one {ms(depth_time[1]):.3f} ms, '
          f'two or more {ms(depth_time[2]):.3f} ms; kernel time per queue {({q: round(ms(v), 3) for q, v in queues.items()})}')
    print(f'{"kernel family":34s} {"alone ms":>9s} {"beside ms":>10s}')
    fams = sorted(set(solo) | set(shared), key=lambda f: -(solo[f] + shared[f]))
    for f in fams[:40]:
        print(f'{f:34s} {ms(solo[f]):9.3f} {ms(shared[f]):10.3f}')
    print(f'{"total":34s} {ms(sum(solo.values())):9.3f} {ms(sum(shared.values())):10.3f}')


if __name__ == '__main__':
    main()
