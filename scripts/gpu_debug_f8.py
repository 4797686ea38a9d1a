import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import orabind, blob
from agarcl_amd import _capi
from lockstep import EngineAsEnv
orabind.build()
z = np.load('tests/golden/f8_cell_eats_cell_2p.npz')
cfg = json.loads(str(z['cfg']))
for name, mk in (('hip', lambda: EngineAsEnv(_capi.BatchedEngine, **cfg)), ('ora', lambda: orabind.OraEnv(**cfg))):
    env = mk(); env.seed(int(z['seed'])); env.reset(True); env.load(z['blob0'])
    a = z['actions'][0]
    env.take_actions(a[:, :2], a[:, 2].astype(np.int32)); r = env.step()
    d = blob.parse(env.dump())
    print(name, 'rewards', r, [ (p['pid'], p['n_cells'], p['cell_mass'].tolist(), p['cell_id'].tolist(), p['cells_eaten']) for p in d['players']])
    if name == 'hip': print('flags', env.flags())
