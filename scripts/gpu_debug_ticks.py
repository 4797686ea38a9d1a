import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import orabind, blob
from agarcl_amd import _capi
orabind.build()
cfg = dict(arena_size=250, num_pellets=500, num_viruses=10, mode=6)
A = 4
eng = _capi.BatchedEngine(A, **cfg)
oras = [orabind.OraEnv(**cfg) for _ in range(A)]
seeds = np.arange(11, 11 + A).astype(np.uint32)
eng.seed(seeds); eng.reset(reset_ids=True)
for o, s in zip(oras, seeds): o.seed(int(s)); o.reset(True)
for a in range(A): assert blob.diff(oras[a].dump(), eng.dump(a)) is None
bad = False
for t in range(12):
    eng.tick(1)
    n, pe, ve = eng.events()
    for a in range(A):
        oras[a].tick()
        ope, ove = oras[a].last_events()
        gpe = pe[a, :n[a, 0]]
        d = blob.diff(oras[a].dump(), eng.dump(a))
        if d or not np.array_equal(ope, gpe):
            bad = True
            do = blob.parse(oras[a].dump()); dg = blob.parse(eng.dump(a))
            print('tick', t, 'arena', a, 'DIFF', d)
            print('  oracle events', ope.tolist(), ove.tolist(), ' gpu events', gpe.tolist(), ve[a, :n[a, 1]].tolist())
            print('  npel', len(do['pellet_x']), len(dg['pellet_x']), 'cells', do['players'][0]['cell_mass'], dg['players'][0]['cell_mass'])
    if bad: break
print('done, bad =', bad)
