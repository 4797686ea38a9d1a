"""TEST INFRASTRUCTURE: host restatement of the reference's GoBiggerObservation::add_frame
(/root/reference/environment/envs/GoBiggerEnvironment.hpp:419-541) from one arena's state blob.  The product's GoBigger path is
the HIP kernel k_gobigger_obs (agarcl_amd/csrc/agar_gobigger.inl) + the tensor -> object view of agarcl_amd/gobigger.py; tests
compare the two.  Parity UNPINNED: GoBiggerEnvironment.hpp does not compile here without OpenGL stand-ins (its constructor
initialises a FrameBufferObject, :582), so this file follows the reference's source text, not its output."""
import numpy as np

from agarcl_amd import snapshot
from agarcl_amd.gobigger import CloneInfo, FoodInfo, Location, SporeInfo, VirusInfo

_f = np.float32


def _radius(mass):                   # core/utils.hpp:8-11: (distance) sqrt(mass / 1.0 / pi) in double, then float
    return float(_f(np.sqrt(np.float64(mass) / 1.0 / np.pi)))


import ctypes as _C
_libm = _C.CDLL("libm.so.6")          # the reference's std::atan(float) is glibc's atanf (numpy's float32 arctan differs by an ulp)
_libm.atanf.restype = _C.c_float; _libm.atanf.argtypes = [_C.c_float]


def _direction(dx, dy):              # Velocity::direction, core/types.hpp:167-174 (atan(dx/dy), not atan2)
    with np.errstate(divide="ignore", invalid="ignore"):
        ang = _f(_libm.atanf(_C.c_float(float(_f(dx) / _f(dy)))))
    if dx < 0:
        ang = _f(np.float64(ang) + np.pi) if dy > 0 else _f(np.float64(ang) - np.pi)
    return float(ang)


def add_frame(player_states, blob_words, grid_size=128):
    """GoBiggerObservation::add_frame (:519-541): refresh the entity lists of EVERY player in the map, in the map's
    iteration order; an entity is listed when it falls inside the player's egocentric grid (:446-514)."""
    d = snapshot.parse_blob(blob_words)
    G = int(grid_size)
    centering = _f(G) / _f(2)

    def f2i(v):  # static_cast<int>(float) on x86-64
        return int(v) if np.isfinite(v) and -2147483904.0 < v < 2147483648.0 else -2147483648

    for pl in d["players"]:
        cells = pl["cells"]
        sx = _f(0); sy = _f(0); tm = 0
        for c in cells:              # Player::x/y/mass (core/Player.hpp:102-126)
            m = int(c[6]); x = np.array([c[0]], np.uint32).view(np.float32)[0]; y = np.array([c[1]], np.uint32).view(np.float32)[0]
            sx = _f(sx + _f(x * _f(m))); sy = _f(sy + _f(y * _f(m))); tm += m
        with np.errstate(divide="ignore", invalid="ignore"):
            px, py = _f(sx / _f(tm)), _f(sy / _f(tm))
        view = _f(min(max(_f(2 * tm), _f(100)), _f(300)))   # clamp<float>(2 * mass, 100, 300), :424-426

        def inside(ex, ey):
            gx = f2i(_f(_f(_f(G) * _f(_f(ex) - px)) / view) + centering); gy = f2i(_f(_f(_f(G) * _f(_f(ey) - py)) / view) + centering)
            return 0 <= gx < G and 0 <= gy < G

        # `auto pstate = player_states.get_player_state(pid)` is a COPY that is cleared and refilled; it replaces the stored
        # state only from inside the `if (_inside_grid)` of _store_entities (:501-504), i.e. when at least one entity is listed
        import copy
        ps = copy.copy(player_states.get_player_state(pl["pid"]))
        ps._food, ps._virus, ps._spore, ps._clone = [], [], [], []
        committed = False
        rel = lambda ex, ey: Location(_f(_f(ex) - px), _f(_f(ey) - py))
        for x, y, m in zip(d["viruses"]["x"], d["viruses"]["y"], d["viruses"]["mass"]):
            if inside(x, y):
                ps._virus.append(VirusInfo(rel(x, y), _radius(int(m)), int(m), (0.0, 0.0))); ps._score = float(tm); committed = True
        for x, y in zip(d["pellets"]["x"], d["pellets"]["y"]):
            if inside(x, y):
                ps._food.append(FoodInfo(rel(x, y), _radius(1), 1)); ps._score = float(tm); committed = True
        for x, y in zip(d["foods"]["x"], d["foods"]["y"]):
            if inside(x, y):
                ps._spore.append(SporeInfo(rel(x, y), _radius(10), 10, (0.0, 0.0), pl["pid"])); ps._score = float(tm); committed = True
        for c in cells:
            x, y, vx, vy = (np.array([c[k]], np.uint32).view(np.float32)[0] for k in range(4))
            if inside(x, y):
                ps._clone.append(CloneInfo(rel(x, y), _radius(int(c[6])), int(c[6]), (float(vx), float(vy)), _direction(vx, vy), pl["pid"], 0)); ps._score = float(tm); committed = True
        if committed:
            player_states.update_player_state(pl["pid"], ps)
    return player_states
