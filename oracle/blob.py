"""State-blob codec (layout: oracle/BLOB_FORMAT.md).  TEST INFRASTRUCTURE.

The same 32-bit-word layout is written by
  * oracle/ref_harness.cpp  (the real reference engine),
  * oracle/agar_oracle.c    (the plain-C restatement),
  * agarcl_amd (C-ABI ``agarcl_dump_arena``; the HIP engine).
so that states can be compared word for word (floats are bit-cast; any-NaN == any-NaN).
"""
import numpy as np

MAGIC = 0x31524741
PLAYER_HDR = 17
CELL_WORDS = 9


def _f(a):
    return np.asarray(a, dtype=np.uint32).view(np.float32)


def parse(blob):
    """blob (uint32 array) -> nested dict of numpy arrays / ints."""
    b = np.asarray(blob, dtype=np.uint32)
    assert b[0] == MAGIC, "bad blob magic"
    out = {"ticks": int(b[1]), "id_counter": int(b[2]), "next_pid": int(b[3])}
    np_, nv, nf, npl = (int(x) for x in b[4:8])
    p = 8
    out["pellet_x"] = _f(b[p:p + np_]); p += np_
    out["pellet_y"] = _f(b[p:p + np_]); p += np_
    out["pellet_id"] = b[p:p + np_].astype(np.int64); p += np_
    for name, isf in (("x", 1), ("y", 1), ("vx", 1), ("vy", 1), ("mass", 0), ("hits", 0), ("id", 0)):
        seg = b[p:p + nv]; p += nv
        out["virus_" + name] = _f(seg) if isf else seg.astype(np.int64)
    for name, isf in (("x", 1), ("y", 1), ("vx", 1), ("vy", 1), ("id", 0)):
        seg = b[p:p + nf]; p += nf
        out["food_" + name] = _f(seg) if isf else seg.astype(np.int64)
    players = []
    for _ in range(npl):
        h = b[p:p + PLAYER_HDR]
        nt = int(h[16])
        pl = {
            "pid": int(h[0]), "is_bot": int(h[1]), "n_cells": int(h[2]), "action": int(h[3]),
            "target": _f(h[4:6]).copy(), "split_cd": int(h[6]), "feed_cd": int(h[7]),
            "elapsed": int(h[8]), "last_decay": int(h[9]), "anti_team": float(_f(h[10:11])[0]),
            "food_eaten": int(h[11]), "highest_mass": int(h[12]), "cells_eaten": int(h[13]),
            "viruses_eaten": int(h[14]), "min_mass_cell": int(h[15]),
            "virus_ticks": b[p + PLAYER_HDR:p + PLAYER_HDR + nt].astype(np.int64),
        }
        p += PLAYER_HDR + nt
        nc = pl["n_cells"]
        cells = b[p:p + nc * CELL_WORDS].reshape(nc, CELL_WORDS); p += nc * CELL_WORDS
        pl["cell_f"] = _f(cells[:, 0:6].copy()).reshape(nc, 6)       # x y vx vy svx svy
        pl["cell_mass"] = cells[:, 6].astype(np.int64)
        pl["cell_id"] = cells[:, 7].astype(np.int64)
        pl["cell_recomb"] = cells[:, 8].astype(np.int64)
        players.append(pl)
    assert p == len(b), "blob length mismatch (%d vs %d)" % (p, len(b))
    out["players"] = players
    return out


def build(d):
    """Inverse of parse()."""
    def fw(a):
        return np.asarray(a, dtype=np.float32).view(np.uint32)

    def iw(a):
        return (np.asarray(a, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32)
    parts = [np.array([MAGIC, d["ticks"], d["id_counter"], d["next_pid"], len(d["pellet_x"]),
                       len(d["virus_x"]), len(d["food_x"]), len(d["players"])], dtype=np.uint32)]
    parts += [fw(d["pellet_x"]), fw(d["pellet_y"]), iw(d["pellet_id"])]
    parts += [fw(d["virus_x"]), fw(d["virus_y"]), fw(d["virus_vx"]), fw(d["virus_vy"]),
              iw(d["virus_mass"]), iw(d["virus_hits"]), iw(d["virus_id"])]
    parts += [fw(d["food_x"]), fw(d["food_y"]), fw(d["food_vx"]), fw(d["food_vy"]), iw(d["food_id"])]
    for pl in d["players"]:
        nc = len(pl["cell_mass"])
        h = np.zeros(PLAYER_HDR, dtype=np.uint32)
        h[0], h[1], h[2], h[3] = pl["pid"], pl["is_bot"], nc, pl["action"]
        h[4:6] = fw(pl["target"])
        h[6], h[7], h[8], h[9] = pl["split_cd"], pl["feed_cd"], pl["elapsed"], pl["last_decay"]
        h[10] = fw([pl["anti_team"]])[0]
        h[11], h[12], h[13], h[14] = pl["food_eaten"], pl["highest_mass"], pl["cells_eaten"], pl["viruses_eaten"]
        h[15], h[16] = pl["min_mass_cell"], len(pl["virus_ticks"])
        parts += [h, iw(pl["virus_ticks"])]
        cells = np.zeros((nc, CELL_WORDS), dtype=np.uint32)
        if nc:
            cells[:, 0:6] = fw(np.asarray(pl["cell_f"], dtype=np.float32).reshape(nc, 6)).reshape(nc, 6)
            cells[:, 6] = iw(pl["cell_mass"]); cells[:, 7] = iw(pl["cell_id"]); cells[:, 8] = iw(pl["cell_recomb"])
        parts.append(cells.reshape(-1))
    return np.concatenate(parts).astype(np.uint32)


def _is_float_word_mask(b):
    """mask of words that hold floats (for NaN-aware / tolerance comparison)."""
    b = np.asarray(b, dtype=np.uint32)
    m = np.zeros(len(b), dtype=bool)
    np_, nv, nf, npl = (int(x) for x in b[4:8])
    p = 8
    m[p:p + 2 * np_] = True; p += 3 * np_
    m[p:p + 4 * nv] = True; p += 7 * nv
    m[p:p + 4 * nf] = True; p += 5 * nf
    for _ in range(npl):
        nc, nt = int(b[p + 2]), int(b[p + 16])
        m[p + 4:p + 6] = True; m[p + 10] = True
        p += PLAYER_HDR + nt
        for _c in range(nc):
            m[p:p + 6] = True; p += CELL_WORDS
    return m


def diff(a, b, rtol=0.0, ignore_velocity=False):
    """Return None if blobs agree, else a short description of the first mismatch.

    rtol == 0 -> bit-exact (NaN == NaN).  rtol > 0 applies to float words only; integer words are
    always exact."""
    a = np.asarray(a, dtype=np.uint32); b = np.asarray(b, dtype=np.uint32)
    if len(a) != len(b):
        return "length %d vs %d (counts %s vs %s)" % (len(a), len(b), a[1:8].tolist(), b[1:8].tolist())
    if np.array_equal(a, b):
        return None
    if not np.array_equal(a[:8], b[:8]):
        return "header %s vs %s" % (a[:8].tolist(), b[:8].tolist())
    fm = _is_float_word_mask(a)
    ne = a != b
    bad_int = ne & ~fm
    if bad_int.any():
        i = int(np.argmax(bad_int))
        return "int word %d: %d vs %d  (%s)" % (i, a[i], b[i], locate(a, i))
    fa, fb = a.view(np.float32), b.view(np.float32)
    both_nan = np.isnan(fa) & np.isnan(fb)
    with np.errstate(invalid="ignore"):
        if rtol > 0:
            close = np.abs(fa - fb) <= rtol * np.maximum(np.abs(fa), np.abs(fb)) + 1e-30
        else:
            close = np.zeros(len(a), dtype=bool)
    bad = ne & fm & ~both_nan & ~close
    if bad.any():
        i = int(np.argmax(bad))
        return "float word %d: %r vs %r (%s)" % (i, float(fa[i]), float(fb[i]), locate(a, i))
    return None


def locate(b, idx):
    """Human-readable location of word idx in blob b."""
    np_, nv, nf, npl = (int(x) for x in b[4:8])
    p = 8
    for name in ("pellet_x", "pellet_y", "pellet_id"):
        if idx < p + np_:
            return "%s[%d]" % (name, idx - p)
        p += np_
    for name in ("x", "y", "vx", "vy", "mass", "hits", "id"):
        if idx < p + nv:
            return "virus_%s[%d]" % (name, idx - p)
        p += nv
    for name in ("x", "y", "vx", "vy", "id"):
        if idx < p + nf:
            return "food_%s[%d]" % (name, idx - p)
        p += nf
    hdr = ["pid", "is_bot", "n_cells", "action", "tx", "ty", "split_cd", "feed_cd", "elapsed", "last_decay",
           "anti_team", "food_eaten", "highest_mass", "cells_eaten", "viruses_eaten", "min_mass_cell", "n_vticks"]
    cw = ["x", "y", "vx", "vy", "svx", "svy", "mass", "id", "recomb"]
    for k in range(npl):
        nc, nt = int(b[p + 2]), int(b[p + 16])
        if idx < p + PLAYER_HDR:
            return "player#%d.%s" % (k, hdr[idx - p])
        p += PLAYER_HDR
        if idx < p + nt:
            return "player#%d.vtick[%d]" % (k, idx - p)
        p += nt
        if idx < p + nc * CELL_WORDS:
            return "player#%d.cell[%d].%s" % (k, (idx - p) // CELL_WORDS, cw[(idx - p) % CELL_WORDS])
        p += nc * CELL_WORDS
    return "?"
