"""GoBigger-style structured observation: the Python-object view of the reference's
GoBiggerObservation (/root/reference/environment/envs/GoBiggerEnvironment.hpp:30-548) with the value classes pybind11
exposes (/root/reference/environment/bindings.cpp:184-318), built on the host from one arena's state.

The entity lists are produced on the GPU as padded tensors (agarcl_gobigger_obs, csrc/agar_gobigger.inl: what a batched learner
consumes); this module turns one arena's rows into the reference's ragged Python objects.
Parity: restated from the reference's source, UNPINNED (GoBiggerEnvironment.hpp does not compile without OpenGL
stand-ins: its constructor initialises a FrameBufferObject, :582)."""
import numpy as np

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


class GoBiggerObservation:           # :251-548 as pybind exposes it, bindings.cpp:297-318 (the module-level class; the env builds its states itself)
    def __init__(self, map_width, map_height, frame_limit, last_frame, team_num):
        self._global = GlobalState(map_width, map_height, frame_limit, last_frame, team_num)
        self._players = PlayerStates({})                  # :301-302: starts empty

    def update_global_state(self, frame_count):           # :344-346
        self._global.update_last_frame_count(frame_count)

    def update_player_state(self, player_id, food_infos, virus_infos, spore_infos, clone_infos, team_name, score, can_eject, can_split):
        # :350-372 (both pybind overloads bind this one function; the second differs in keyword names only, accepted below)
        self._players.update_player_state(player_id, PlayerState(player_id, food_infos, virus_infos, spore_infos, clone_infos, team_name, score, can_eject, can_split))
        print("Updated player state for player ID: %d" % int(player_id))   # :370: the reference prints this line

    def get_global_state(self):                           # :374
        return self._global

    def get_player_states(self):                          # :375
        return self._players


def _second_overload(fn):
    """bindings.cpp:312-315: the same function under the keyword names food_positions / thorn_positions / spore_positions / clone_positions"""
    def update_player_state(self, *a, **kw):
        for new, old in (("food_positions", "food_infos"), ("thorn_positions", "virus_infos"), ("spore_positions", "spore_infos"), ("clone_positions", "clone_infos")):
            if new in kw:
                kw[old] = kw.pop(new)
        return fn(self, *a, **kw)
    update_player_state.__doc__ = fn.__doc__
    return update_player_state


GoBiggerObservation.update_player_state = _second_overload(GoBiggerObservation.update_player_state)


def add_frame(player_states, tensors, arena=0):
    """GoBiggerObservation::add_frame (:519-541) for one arena from the tensors of BatchedEngine.gobigger_obs(): the entity
    lists of every player whose row is marked committed are replaced (the reference commits a refreshed state only when at
    least one entity lies inside the player's grid, :501-504); the others keep their previous state."""
    hdr = tensors["hdr"][arena]
    for k in range(hdr.shape[0]):
        pid, committed, nv, nf, ns, nc, score = (int(v) for v in hdr[k, :7])
        ps = player_states.get_player_state(pid)          # creates the "dummy" state on first use (:227-243)
        if not committed:
            continue
        V, F, S, Cn = tensors["virus"][arena, k], tensors["food"][arena, k], tensors["spore"][arena, k], tensors["clone"][arena, k]
        ps._virus = [VirusInfo(Location(r[0], r[1]), r[2], int(r[3]), (0.0, 0.0)) for r in V[:min(nv, len(V))]]
        ps._food = [FoodInfo(Location(r[0], r[1]), r[2], int(r[3])) for r in F[:min(nf, len(F))]]
        ps._spore = [SporeInfo(Location(r[0], r[1]), r[2], int(r[3]), (0.0, 0.0), pid) for r in S[:min(ns, len(S))]]
        ps._clone = [CloneInfo(Location(r[0], r[1]), r[2], int(r[3]), (float(r[4]), float(r[5])), r[6], pid, 0) for r in Cn[:min(nc, len(Cn))]]
        ps._score = float(score)
    return player_states
