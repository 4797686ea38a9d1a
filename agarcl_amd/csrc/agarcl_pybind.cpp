// Python module `agarcl`: the reference's pybind11 surface (/root/reference/environment/bindings.cpp:94-376) bound to the
// HIP engine through the C ABI of include/agarcl_batch.h (num_arenas == 1 per object, like the reference's classes).
//
// Host-only translation unit (g++, no HIP headers): it links against libagarcl_hip.so and calls nothing but the C ABI.
// Class names, positional constructor signatures, method names, return types and error behaviour (C++ exceptions ->
// Python RuntimeError) follow bindings.cpp; what each method replaces is cited next to it.  Two pieces stay on the
// Python side because they are host text / host objects by nature and already exist there: the JSON snapshot codec
// (agarcl_amd/snapshot.py: the reference's wire format) and the GoBigger value classes built from the kernel's tensors
// (agarcl_amd/gobigger.py).  Built by agarcl_amd/build.py:build_pybind() into <repo>/agarcl.<abi>.so.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/agarcl_batch.h"

namespace py = pybind11;

namespace {

void check(int rc) {  // EngineException / EnvironmentException -> RuntimeError (Engine.hpp:25-27, BaseEnvironment.hpp:19-21)
  if (rc != 0) throw std::runtime_error(std::string("agarcl error ") + std::to_string(rc) + ": " + agarcl_last_error());
}

// The operations agarcl_amd/snapshot.py needs from an engine object (dump / adopt / seeds / arena words), over the C ABI.
struct EngineProxy {
  agarcl_env *h = nullptr;
  int players() const { return agarcl_players_per_arena(h); }
  py::array_t<uint32_t> dump(int arena) {
    std::vector<uint32_t> buf(1 << 16);
    int n = agarcl_dump_arena(h, arena, buf.data(), (int)buf.size());
    if (n < -1000) { buf.resize((size_t)(-n)); n = agarcl_dump_arena(h, arena, buf.data(), (int)buf.size()); }
    if (n < 0) check(n);
    return py::array_t<uint32_t>((py::ssize_t)n, buf.data());
  }
  void adopt(int arena, py::array_t<uint32_t, py::array::c_style | py::array::forcecast> blob, py::array_t<int32_t, py::array::c_style | py::array::forcecast> kinds, int hb, int hr) {
    check(agarcl_adopt_arena(h, arena, blob.data(), (int32_t)blob.size(), kinds.data(), hb, hr));
  }
  void seed_arena(int arena, uint64_t seed) { check(agarcl_seed_arena(h, arena, (uint32_t)(seed & 0xFFFFFFFFu))); }
  py::array_t<uint32_t> seeds() {
    py::array_t<uint32_t> out((py::ssize_t)agarcl_num_arenas(h));
    check(agarcl_get_seeds(h, out.mutable_data()));
    return out;
  }
  py::tuple arena_words(int arena) {
    py::array_t<int32_t> ar(AGARCL_ARENA_WORDS), pl({(py::ssize_t)players(), (py::ssize_t)AGARCL_PLAYER_WORDS});
    check(agarcl_get_arena_words(h, arena, ar.mutable_data(), pl.mutable_data()));
    return py::make_tuple(ar, pl);
  }
};

struct Environment {  // BaseEnvironment's Python-visible surface
  agarcl_env *env = nullptr;
  int n_agents = 0;
  bool loaded = false;       // BaseEnvironment::is_loading_env_state: reset() is a no-op once a snapshot is in (:180-181)
  py::object names = py::none();
  py::dict cfg;              // what save_env_state writes as configuration (BaseEnvironment.hpp:213-231)

  Environment(int num_agents, int ticks_per_step, int arena_size, bool pellet_regen, int num_pellets, int num_viruses,
              int num_bots, int reward_type, int c_death, int mode_number, bool screen_respawn) : n_agents(num_agents) {
    agarcl_config c{};
    c.num_agents = num_agents; c.ticks_per_step = ticks_per_step; c.arena_size = arena_size; c.pellet_regen = pellet_regen;
    c.num_pellets = num_pellets; c.num_viruses = num_viruses; c.num_bots = num_bots; c.reward_type = reward_type;
    c.c_death = c_death; c.mode_number = mode_number; c.screen_respawn = screen_respawn;
    check(agarcl_create(&c, /*num_arenas=*/1, /*device=*/0, &env));
    cfg["num_agents"] = num_agents; cfg["ticks_per_step"] = ticks_per_step; cfg["arena_size"] = arena_size; cfg["num_bots"] = num_bots;
    cfg["reward_type"] = reward_type != 0; cfg["c_death"] = c_death; cfg["pellet_regen"] = pellet_regen;
    cfg["mode_number"] = 0;  // Engine::mode_number as the reference's writer sees it (Engine.hpp:354): 0 until a snapshot is loaded
  }
  Environment(const Environment &) = delete;
  Environment &operator=(const Environment &) = delete;
  virtual ~Environment() { close(); }

  void seed(int s) { uint32_t v = (uint32_t)s; check(agarcl_seed(env, &v, 0)); }               // bindings.cpp:103
  virtual void reset() { if (!loaded) check(agarcl_reset(env, nullptr, 0)); }                    // bindings.cpp:130
  void take_actions(const py::list &actions) {                                                   // bindings.cpp:50-64,117-119
    if ((int)actions.size() != n_agents)                                                         // BaseEnvironment.hpp:142-144
      throw std::runtime_error("Number of actions (" + std::to_string(actions.size()) + ") does not match number of agents (" + std::to_string(n_agents) + ")");
    std::vector<float> dxdy; std::vector<int32_t> act;
    for (auto &a : actions) {
      auto t = py::cast<py::tuple>(a);
      dxdy.push_back(py::cast<float>(t[0])); dxdy.push_back(py::cast<float>(t[1])); act.push_back(py::cast<int>(t[2]));
    }
    check(agarcl_set_actions(env, dxdy.data(), act.data(), /*on_device=*/0));
  }
  virtual std::vector<double> step() {                                                           // bindings.cpp:132
    check(agarcl_step(env, 0));
    std::vector<double> r((size_t)n_agents);
    check(agarcl_get_rewards(env, r.data()));
    return r;
  }
  std::vector<bool> dones() {                                                                    // bindings.cpp:116
    std::vector<uint8_t> d((size_t)n_agents);
    check(agarcl_get_dones(env, d.data()));
    return std::vector<bool>(d.begin(), d.end());
  }
  void render() {}
  void close() { if (env) { agarcl_destroy(env); env = nullptr; } }
  EngineProxy proxy() { EngineProxy p; p.h = env; return p; }
  void save_env_state(const std::string &path) {                                                 // bindings.cpp:131 -> BaseEnvironment.hpp:213-310
    py::module_ snap = py::module_::import("agarcl_amd.snapshot");
    py::object s = snap.attr("save_arena")(proxy(), 0, cfg, names);
    py::object text = snap.attr("dumps")(s);
    py::object f;
    try { f = py::module_::import("builtins").attr("open")(path, "w"); }
    catch (py::error_already_set &) { throw std::runtime_error("Failed to open " + path + " for writing"); }   // BaseEnvironment.hpp:304-306
    f.attr("write")(text); f.attr("close")();
  }
  void load_env_state(const std::string &path) {                                                 // bindings.cpp:170,372 -> BaseEnvironment.hpp:312-343
    py::object f;
    try { f = py::module_::import("builtins").attr("open")(path); }
    catch (py::error_already_set &) { throw std::runtime_error("Failed to open " + path + " for reading"); }   // Engine.hpp:255-257
    py::object snapd = py::module_::import("json").attr("load")(f);
    f.attr("close")();
    names = py::module_::import("agarcl_amd.snapshot").attr("load_arena")(proxy(), 0, snapd, py::arg("reset_ids") = false);
    cfg["mode_number"] = snapd["mode_number"];                                                   // Engine.hpp:263
    loaded = true;
  }
};

struct GridEnvironment : Environment {  // agario::env::GridEnvironment<int, renderable>, bindings.cpp:99-135
  bool configured = false;
  int num_frames = 1, grid_size = 128, ticks_per_step_; bool cells = true, others = true, viruses = true, pellets = true, literal = false;
  GridEnvironment(int num_agents, int ticks_per_step, int arena_size, bool pellet_regen, int num_pellets, int num_viruses,
                  int num_bots, int reward_type, int /*c_death*/, int mode_number)
      // GridEnvironment forwards c_death = 0 to its base (GridEnvironment.hpp:369-372)
      : Environment(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, 0, mode_number, false), ticks_per_step_(ticks_per_step) {}
  void configure_observation(const py::dict &config) {                                           // bindings.cpp:104-114
    num_frames = config.contains("num_frames") ? config["num_frames"].cast<int>() : 1;
    grid_size = config.contains("grid_size") ? config["grid_size"].cast<int>() : 128;
    cells = config.contains("observe_cells") ? config["observe_cells"].cast<bool>() : true;
    others = config.contains("observe_others") ? config["observe_others"].cast<bool>() : true;
    viruses = config.contains("observe_viruses") ? config["observe_viruses"].cast<bool>() : true;
    pellets = config.contains("observe_pellets") ? config["observe_pellets"].cast<bool>() : true;
    // extra key (the reference ignores unknown keys): place the frame exactly where the reference's arithmetic puts it
    literal = config.contains("literal_frame_index") ? config["literal_frame_index"].cast<bool>() : false;
    if (num_frames < 1) throw std::runtime_error("num_frames must be positive");
    configured = true;
  }
  int frame_channels() const { return 1 + cells + 2 * others + 2 * viruses + 2 * pellets; }
  py::tuple observation_shape() {                                                                // bindings.cpp:115 (GridEnvironment.hpp:72-88)
    if (!configured) throw std::runtime_error("GridObservation was not configured.");
    return py::make_tuple(num_frames * frame_channels(), grid_size, grid_size);
  }
  py::list get_state() {  // bindings.cpp:67-91,133: one OWNED int32 (C, G, G) array per agent
    if (!configured) throw std::runtime_error("GridObservation was not configured.");
    const int C = frame_channels(), G = grid_size;
    std::vector<int32_t> buf((size_t)n_agents * C * G * G);
    int ch = 0;
    check(agarcl_grid_obs(env, G, cells, others, viruses, pellets, buf.data(), /*on_device=*/0, &ch));
    py::list obs;
    for (int i = 0; i < n_agents; i++) {
      // The observation holds num_frames frame slots, cleared at every step (GridEnvironment.hpp:91-123,405-410).  The
      // reference calls _partial_observation(agent, tick_index = 0) once per step and stores the frame at
      // frame_index = 0 - (ticks_per_step - num_frames) if that is >= 0 (:417-431): with the default arguments (1 frame,
      // 4 ticks) NO frame is ever stored and the observation is all zeros.  Default here = the evident intent: the state
      // after the step in the LAST slot; literal_frame_index = true reproduces the reference's output exactly.
      py::array_t<int32_t> a({(py::ssize_t)(num_frames * C), (py::ssize_t)G, (py::ssize_t)G});
      int32_t *d = a.mutable_data();
      std::fill(d, d + (size_t)num_frames * C * G * G, 0);
      const int slot = literal ? num_frames - ticks_per_step_ : num_frames - 1;
      if (slot >= 0 && slot < num_frames) std::copy(buf.begin() + (size_t)i * C * G * G, buf.begin() + (size_t)(i + 1) * C * G * G, d + (size_t)slot * C * G * G);
      obs.append(a);
    }
    return obs;
  }
  py::array_t<uint8_t> get_frame() {                                                             // bindings.cpp:120-129: uint8 (1, 512, 512, 3)
    py::array_t<uint8_t> all({(py::ssize_t)n_agents, (py::ssize_t)512, (py::ssize_t)512, (py::ssize_t)3});
    check(agarcl_screen_obs(env, 512, 512, 0, all.mutable_data(), 0));
    py::array_t<uint8_t> out({(py::ssize_t)1, (py::ssize_t)512, (py::ssize_t)512, (py::ssize_t)3});
    std::copy(all.data() + (size_t)(n_agents - 1) * 512 * 512 * 3, all.data() + (size_t)n_agents * 512 * 512 * 3, out.mutable_data());
    return out;
  }
};

struct ScreenEnvironment : Environment {  // agario::env::ScreenEnvironment<renderable>, bindings.cpp:142-171
  int w, h; bool agent_view;
  ScreenEnvironment(int num_agents, int frames_per_step, int arena_size, bool pellet_regen, int num_pellets, int num_viruses, int num_bots,
                    bool reward_type, int c_death, int mode_number, bool load_env_snapshot, int screen_width, int screen_height, bool agent_view_)
      // the respawn hook of ScreenEnvironment::_partial_observation (ScreenEnvironment.hpp:233-243) is agarcl_config.screen_respawn
      : Environment(num_agents, frames_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, c_death, mode_number, true),
        w(screen_width), h(screen_height), agent_view(agent_view_) { loaded = load_env_snapshot; }
  py::tuple observation_shape() { return py::make_tuple(1, w, h, agent_view ? 4 : 3); }          // bindings.cpp:156
  py::list get_state() {  // bindings.cpp:157-168: a list holding the ONE frame buffer, uint8 (1, W, H, 3|4)
    const int c = agent_view ? 4 : 3;
    py::array_t<uint8_t> all({(py::ssize_t)n_agents, (py::ssize_t)h, (py::ssize_t)w, (py::ssize_t)c});
    check(agarcl_screen_obs(env, w, h, agent_view, all.mutable_data(), 0));
    // every agent's render overwrites the one buffer in turn: the last agent's frame remains
    py::array_t<uint8_t> out({(py::ssize_t)1, (py::ssize_t)w, (py::ssize_t)h, (py::ssize_t)c});
    std::copy(all.data() + (size_t)(n_agents - 1) * h * w * c, all.data() + (size_t)n_agents * h * w * c, out.mutable_data());
    py::list obs; obs.append(out);
    return obs;
  }
};

struct GoBiggerEnvironment : Environment {  // agario::env::GoBiggerEnvironment<renderable>, bindings.cpp:321-375
  py::object gb, global_state, player_states;
  int grid_size = 128, no_frames = 0;
  GoBiggerEnvironment(int map_width, int map_height, int frame_limit, int num_agents, int ticks_per_step, int arena_size, bool pellet_regen,
                      int num_pellets, int num_viruses, int num_bots, bool reward_type, int /*c_death*/, int mode_number, bool load_env_snapshot, bool /*agent_view*/)
      // _partial_observation zeroes c_death_ on every call (GoBiggerEnvironment.hpp:624): death is never penalised here
      : Environment(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, 0, mode_number, false) {
    gb = py::module_::import("agarcl_amd.gobigger");
    global_state = gb.attr("GlobalState")(map_width, map_height, frame_limit, 0, num_agents);
    player_states = gb.attr("PlayerStates")();
    loaded = load_env_snapshot;
  }
  void configure_observation(const py::dict &config) { grid_size = config.contains("grid_size") ? config["grid_size"].cast<int>() : 128; }   // bindings.cpp:341-352
  void observe() {  // _partial_observation per agent -> GoBiggerObservation::add_frame (:519-541, 618-636), from the kernel's padded tensors
    const int P = agarcl_players_per_arena(env), KF = 256, KV = 64, KS = 64, KC = 32;
    py::array_t<int32_t> hdr({1, P, 8});
    py::array_t<float> food({1, P, KF, 4}), virus({1, P, KV, 4}), spore({1, P, KS, 4}), clone({1, P, KC, 7});
    check(agarcl_gobigger_obs(env, grid_size, KF, KV, KS, KC, hdr.mutable_data(), food.mutable_data(), virus.mutable_data(), spore.mutable_data(), clone.mutable_data(), 0));
    py::dict t; t["hdr"] = hdr; t["food"] = food; t["virus"] = virus; t["spore"] = spore; t["clone"] = clone;
    gb.attr("add_frame")(player_states, t, 0);
    no_frames += n_agents;
    global_state.attr("update_last_frame_count")(0);
  }
  void reset() override { Environment::reset(); if (!loaded) observe(); }
  std::vector<double> step() override { auto r = Environment::step(); observe(); return r; }
  py::tuple observation_shape() { return py::make_tuple(no_frames, global_state.attr("get_map_height")(), global_state.attr("get_map_width")()); }   // :411-416
  py::list get_state() { py::dict d; d["global_state"] = global_state; d["player_states"] = player_states; py::list l; l.append(d); return l; }      // bindings.cpp:28-47
  py::array_t<uint8_t> get_frame() {                                                             // bindings.cpp:354-363
    py::array_t<uint8_t> all({(py::ssize_t)n_agents, (py::ssize_t)512, (py::ssize_t)512, (py::ssize_t)3});
    check(agarcl_screen_obs(env, 512, 512, 0, all.mutable_data(), 0));
    py::array_t<uint8_t> out({(py::ssize_t)1, (py::ssize_t)512, (py::ssize_t)512, (py::ssize_t)3});
    std::copy(all.data() + (size_t)(n_agents - 1) * 512 * 512 * 3, all.data() + (size_t)n_agents * 512 * 512 * 3, out.mutable_data());
    return out;
  }
};

}  // namespace

PYBIND11_MODULE(agarcl, module) {
  module.doc() = "Agar.io Learning Environment -- MI355X-native batched engine behind the reference's `agarcl` surface";

  py::class_<EngineProxy>(module, "_EngineProxy")
      .def_property_readonly("players", &EngineProxy::players)
      .def("dump", &EngineProxy::dump).def("adopt", &EngineProxy::adopt).def("seed_arena", &EngineProxy::seed_arena)
      .def("seeds", &EngineProxy::seeds).def("arena_words", &EngineProxy::arena_words);

  py::class_<GridEnvironment>(module, "GridEnvironment")
      .def(py::init<int, int, int, bool, int, int, int, int, int, int>())
      .def("seed", &GridEnvironment::seed)
      .def("configure_observation", &GridEnvironment::configure_observation)
      .def("observation_shape", &GridEnvironment::observation_shape)
      .def("dones", &GridEnvironment::dones)
      .def("take_actions", &GridEnvironment::take_actions)
      .def("get_frame", &GridEnvironment::get_frame)
      .def("reset", &GridEnvironment::reset)
      .def("render", &GridEnvironment::render)
      .def("step", &GridEnvironment::step)
      .def("get_state", &GridEnvironment::get_state)
      .def("close", &GridEnvironment::close)
      .def("save_env_state", &GridEnvironment::save_env_state);

  py::class_<ScreenEnvironment>(module, "ScreenEnvironment")
      .def(py::init<int, int, int, bool, int, int, int, bool, int, int, bool, int, int, bool>())
      .def("seed", &ScreenEnvironment::seed)
      .def("observation_shape", &ScreenEnvironment::observation_shape)
      .def("dones", &ScreenEnvironment::dones)
      .def("take_actions", &ScreenEnvironment::take_actions)
      .def("reset", &ScreenEnvironment::reset)
      .def("render", &ScreenEnvironment::render)
      .def("step", &ScreenEnvironment::step)
      .def("get_state", &ScreenEnvironment::get_state)
      .def("close", &ScreenEnvironment::close)
      .def("load_env_state", &ScreenEnvironment::load_env_state)
      .def("save_env_state", &ScreenEnvironment::save_env_state);
  module.attr("has_screen_env") = py::bool_(true);   // frames come from the rule-based HIP rasteriser, not from OpenGL

  // GoBigger value classes and GoBiggerObservation (bindings.cpp:184-318): the Python classes of agarcl_amd/gobigger.py under the reference's names
  py::module_ gb = py::module_::import("agarcl_amd.gobigger");
  for (const char *n : {"FoodInfo", "VirusInfo", "SporeInfo", "CloneInfo", "GlobalState", "PlayerState", "PlayerStates", "GoBiggerObservation"}) module.attr(n) = gb.attr(n);

  py::class_<GoBiggerEnvironment>(module, "GoBiggerEnvironment")
      .def(py::init<int, int, int, int, int, int, bool, int, int, int, bool, int, int, bool, bool>(),
           py::arg("map_width"), py::arg("map_height"), py::arg("frame_limit"), py::arg("num_agents"), py::arg("ticks_per_step"),
           py::arg("arena_size"), py::arg("pellet_regen"), py::arg("num_pellets"), py::arg("num_viruses"), py::arg("num_bots"),
           py::arg("reward_type"), py::arg("c_death") = 0, py::arg("mode_number") = 0, py::arg("load_env_snapshot") = false,
           py::arg("agent_view") = false)
      .def("configure_observation", &GoBiggerEnvironment::configure_observation)
      .def("get_state", &GoBiggerEnvironment::get_state)
      .def("get_frame", &GoBiggerEnvironment::get_frame)
      .def("take_actions", &GoBiggerEnvironment::take_actions)
      .def("dones", &GoBiggerEnvironment::dones)
      .def("observation_shape", &GoBiggerEnvironment::observation_shape)
      .def("seed", &GoBiggerEnvironment::seed, "Seed the environment")
      .def("reset", &GoBiggerEnvironment::reset, "Reset the environment")
      .def("step", &GoBiggerEnvironment::step, "Step through the environment")
      .def("render", &GoBiggerEnvironment::render, "Render the current state")
      .def("close", &GoBiggerEnvironment::close, "Close the environment")
      .def("load_env_state", &GoBiggerEnvironment::load_env_state)
      .def("save_env_state", &GoBiggerEnvironment::save_env_state);
}
