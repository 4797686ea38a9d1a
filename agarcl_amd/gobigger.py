"""GoBigger-style structured observation: the Python-object view of the reference's
GoBiggerObservation (/root/reference/environment/envs/GoBiggerEnvironment.hpp:30-548) with the value classes pybind11
exposes (/root/reference/environment/bindings.cpp:184-318), built on the host from one arena's state.

The observation is a ragged dictionary of Python objects -- host data by nature -- so it is derived from the arena's
state blob (agarcl_dump_arena) rather than by a kernel; a padded-tensor form for batched training is not provided.
Parity: restated from the reference's source, UNPINNED (GoBiggerEnvironment.hpp does not compile without OpenGL
stand-ins: its constructor initialises a FrameBufferObject, :582)."""
import numpy as np

from . import snapshot

_f = np.float32


class Location:                      # agario::Location as pybind exposes it through the *Info.position members
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)

    def __repr__(self):
        return "Location(%r, %r)" % (self.x, self.y)


class _Info:
    def get_position_x(self):        # bindings.cpp:186-187
        return float(_f(self.position.x))

    def get_position_y(self):
        return float(_f(self.position.y))


class FoodInfo(_Info):               # GoBiggerEnvironment.hpp:73-78
    def __init__(self, position, radius, score):
        self.position, self.radius, self.score = position, float(radius), int(score)


class VirusInfo(_Info):              # :80-86
    def __init__(self, position, radius, score, velocity):
        self.position, self.radius, self.score, self.velocity = position, float(radius), int(score), velocity


class SporeInfo(_Info):              # :88-95
    def __init__(self, position, radius, score, velocity, owner):
        self.position, self.radius, self.score, self.velocity, self.owner = position, float(radius), int(score), velocity, int(owner)


class CloneInfo(_Info):              # :98-107
    def __init__(self, position, radius, score, velocity, direction, owner, teamId):
        self.position, self.radius, self.score, self.velocity = position, float(radius), int(score), velocity
        self.direction, self.owner, self.teamId = float(direction), int(owner), int(teamId)


class GlobalState:                   # :30-71, bindings.cpp:222-240
    def __init__(self, width, height, frame_limit, last_frame, team_num):
        self._w, self._h, self._fl, self._last, self._teams = int(width), int(height), int(frame_limit), int(last_frame), int(team_num)

    def update_last_frame_count(self, n):
        self._last = int(n)

    def get_map_width(self):
        return self._w

    def get_map_height(self):
        return self._h

    def get_frame_limit(self):
        return self._fl

    def get_team_num(self):
        return self._teams

    def __str__(self):
        return "GlobalState(map_width=%d, map_height=%d, frame_limit=%d, team_num=%d)" % (self._w, self._h, self._fl, self._teams)


class PlayerState:                   # :109-213, bindings.cpp:243-272
    def __init__(self, player_id, food_infos, virus_infos, spore_infos, clone_infos, team_name, score, can_eject, can_split):
        self._id, self._food, self._virus, self._spore, self._clone = int(player_id), list(food_infos), list(virus_infos), list(spore_infos), list(clone_infos)
        self._team, self._score, self._eject, self._split = team_name, float(score), bool(can_eject), bool(can_split)

    def get_player_id(self):
        return self._id

    def get_food_infos(self):
        return self._food

    def get_virus_infos(self):
        return self._virus

    def get_spore_infos(self):
        return self._spore

    def get_clone_infos(self):
        return self._clone

    def get_team_name(self):
        return self._team

    def get_score(self):
        return self._score

    def canEject(self):
        return self._eject

    def canSplit(self):
        return self._split

    def update_score(self, s):
        self._score = float(s)


class PlayerStates:                  # :216-259, bindings.cpp:275-296
    def __init__(self, player_states=None):
        self._m = dict(player_states or {})

    def update_player_state(self, pid, ps):
        self._m[int(pid)] = ps

    def get_player_state(self, pid):
        if int(pid) not in self._m:  # :227-243: a "dummy" state is created on first use
            self._m[int(pid)] = PlayerState(pid, [], [], [], [], "dummy", 0.0, True, True)
        return self._m[int(pid)]

    def get_all_player_states(self):
        return self._m

    def __str__(self):
        out = "PlayerStates:\n"
        for st in self._m.values():
            out += "  Player %d: score=%s, food_seen=%d, virus_seen=%d, spores_seen=%d, no_clone=%d, team_name=\"%s\"\n" % (
                st.get_player_id(), st.get_score(), len(st._food), len(st._virus), len(st._spore), len(st._clone), st.get_team_name())
        return out


def _radius(mass):                   # core/utils.hpp:8-11: (distance) sqrt(mass / 1.0 / pi) in double, then float
    return float(_f(np.sqrt(np.float64(mass) / 1.0 / np.pi)))


def _direction(dx, dy):              # Velocity::direction, core/types.hpp:167-174 (atan(dx/dy), not atan2)
    with np.errstate(divide="ignore", invalid="ignore"):
        ang = np.arctan(_f(dx) / _f(dy), dtype=np.float32)
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

        ps = player_states.get_player_state(pl["pid"])
        ps._food, ps._virus, ps._spore, ps._clone = [], [], [], []
        rel = lambda ex, ey: Location(_f(_f(ex) - px), _f(_f(ey) - py))
        for x, y, m in zip(d["viruses"]["x"], d["viruses"]["y"], d["viruses"]["mass"]):
            if inside(x, y):
                ps._virus.append(VirusInfo(rel(x, y), _radius(int(m)), int(m), (0.0, 0.0))); ps._score = float(tm)
        for x, y in zip(d["pellets"]["x"], d["pellets"]["y"]):
            if inside(x, y):
                ps._food.append(FoodInfo(rel(x, y), _radius(1), 1)); ps._score = float(tm)
        for x, y in zip(d["foods"]["x"], d["foods"]["y"]):
            if inside(x, y):
                ps._spore.append(SporeInfo(rel(x, y), _radius(10), 10, (0.0, 0.0), pl["pid"])); ps._score = float(tm)
        for c in cells:
            x, y, vx, vy = (np.array([c[k]], np.uint32).view(np.float32)[0] for k in range(4))
            if inside(x, y):
                ps._clone.append(CloneInfo(rel(x, y), _radius(int(c[6])), int(c[6]), (float(vx), float(vy)), _direction(vx, vy), pl["pid"], 0)); ps._score = float(tm)
    return player_states
