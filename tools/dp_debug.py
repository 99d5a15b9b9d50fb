"""Developer aid: run one scenario of tests/test_distributed_gpu.py and print the per-parameter gradient error of
the 2-rank run against the single-process run."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp
from tests import test_distributed_gpu as T


def main():
    scenario = sys.argv[1] if len(sys.argv) > 1 else 'pretrain'
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=T._worker, args=(r, world, port, q, scenario, 'gloo')) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    if res[0] != 'ok':
        print(res[1])
        return
    snaps = res[1]
    tr, m, ref = T._drive(scenario, 1, 0)
    ar = m.arena()
    for i, ((g, n), (gr, nr)) in enumerate(zip(snaps, ref)):
        g, gr = torch.as_tensor(g), torch.as_tensor(gr)
        print('step', i, 'norm', n, nr, 'total err', float((g - gr).norm() / gr.norm()))
        for name, (o, cnt, shape) in ar.offsets.items():
            a, b = g[o:o + cnt], gr[o:o + cnt]
            if float(b.norm()) == 0 and float(a.norm()) == 0:
                continue
            e = float((a - b).norm() / (b.norm() + 1e-30))
            if e > 1e-2:
                print('   %-50s err %.3e  |dp| %.3e |ref| %.3e ratio %.3f' % (name, e, float(a.norm()), float(b.norm()), float(a.norm() / (b.norm() + 1e-30))))


if __name__ == '__main__':
    main()
