"""TEST INFRASTRUCTURE.  Host restatement of the "ram" observation (agarcl_amd/csrc/agar_ram.inl, include/agarcl_batch.h agarcl_ram_obs)
from a state blob (oracle/BLOB_FORMAT.md).  There is no reference implementation of a ram observation to pin this against (the reference
rejects obs_type "ram": gym_agario/AgarioEnv.py:211; SURVEY 8d C1 excludes it from parity): PARITY UNPINNED by construction -- this file
only checks that the HIP kernel computes the layout the header documents.  Only tests/ import it."""
import numpy as np

from . import blob

F = np.float32


def _nearest(xs, ys, px, py, k):
    """indices of the k entities nearest to (px, py): key = (bits of the fp32 squared distance, index)"""
    if len(xs) == 0 or k == 0:
        return []
    with np.errstate(invalid="ignore", over="ignore"):
        dx = (np.asarray(xs, F) - F(px)).astype(F); dy = (np.asarray(ys, F) - F(py)).astype(F)
        d = ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F)
    bits = d.view(np.uint32).astype(np.uint64)
    order = np.argsort((bits << np.uint64(32)) | np.arange(len(xs), dtype=np.uint64), kind="stable")
    return [int(i) for i in order[:k]]


def ram_obs(state_blob, agent_slot_pids, k_cells=16, k_pellets=16, k_viruses=8, k_others=16):
    """state_blob: one arena; agent_slot_pids: pid of every agent in agent (= slot) order, followed by the pids of the remaining player slots
    in slot order (the blob lists players in map order; slots are what the kernel walks).  Returns float32 [n_agents, D]."""
    d = blob.parse(state_blob)
    by_pid = {int(p["pid"]): p for p in d["players"]}
    slots = [by_pid[int(pid)] for pid in agent_slot_pids]
    n_agents = sum(1 for p in slots if not p["is_bot"])
    D = 4 + 3 * k_cells + 2 * k_pellets + 3 * k_viruses + 3 * k_others
    out = np.zeros((n_agents, D), F)
    for a in range(n_agents):
        pl = slots[a]
        sx = F(0); sy = F(0); tm = 0
        for c, m in zip(pl["cell_f"], pl["cell_mass"]):
            fm = F(int(m)); sx = F(sx + F(F(c[0]) * fm)); sy = F(sy + F(F(c[1]) * fm)); tm += int(m)
        with np.errstate(invalid="ignore", divide="ignore"):
            px = F(sx) / F(tm); py = F(sy) / F(tm)
        row = out[a]
        if tm == 0:      # a dead agent: the record stays all-zero with cell count 0 (include/agarcl_batch.h)
            continue
        row[0], row[1], row[2], row[3] = px, py, F(tm), F(len(pl["cell_mass"]))
        o = 4
        with np.errstate(invalid="ignore"):
            for i, (c, m) in enumerate(zip(pl["cell_f"], pl["cell_mass"])):
                if i < k_cells:
                    row[o + 3 * i] = F(c[0]) - px; row[o + 3 * i + 1] = F(c[1]) - py; row[o + 3 * i + 2] = F(int(m))
            o += 3 * k_cells
            for r, i in enumerate(_nearest(d["pellet_x"], d["pellet_y"], px, py, k_pellets)):
                row[o + 2 * r] = F(d["pellet_x"][i]) - px; row[o + 2 * r + 1] = F(d["pellet_y"][i]) - py
            o += 2 * k_pellets
            for r, i in enumerate(_nearest(d["virus_x"], d["virus_y"], px, py, k_viruses)):
                row[o + 3 * r] = F(d["virus_x"][i]) - px; row[o + 3 * r + 1] = F(d["virus_y"][i]) - py; row[o + 3 * r + 2] = F(int(d["virus_mass"][i]))
            o += 3 * k_viruses
            xs, ys, ms = [], [], []
            for s, other in enumerate(slots):      # entity order of the kernel: slot, then cell
                if s == a:
                    continue
                for c, m in zip(other["cell_f"], other["cell_mass"]):
                    xs.append(c[0]); ys.append(c[1]); ms.append(int(m))
            for r, i in enumerate(_nearest(xs, ys, px, py, k_others)):
                row[o + 3 * r] = F(xs[i]) - px; row[o + 3 * r + 1] = F(ys[i]) - py; row[o + 3 * r + 2] = F(ms[i])
    return out
