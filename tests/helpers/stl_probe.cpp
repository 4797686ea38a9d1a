// Real libstdc++ / glibc behaviour that the oracle restates (NOT reference code): unordered_map
// iteration order, std::sort on (float key) pairs, mt19937_64 + uniform_real_distribution<float>, rand().
#include <algorithm>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>
extern "C" {
// persistent map so that clear() keeps the bucket array like GameState::clear() does
static std::unordered_map<unsigned short, int> *g_map = nullptr;
void probe_map_new() { delete g_map; g_map = new std::unordered_map<unsigned short, int>(); }
int probe_map_round(const int *keys, int n, int *order_out) {
  g_map->clear();
  for (int i = 0; i < n; i++) g_map->insert(std::make_pair((unsigned short)keys[i], i));
  int k = 0;
  for (auto &kv : *g_map) order_out[k++] = kv.first;
  return k;
}
int probe_intmap_order(const int *keys, int n, int *order_out) {
  std::unordered_map<int, std::vector<int>> m;
  for (int i = 0; i < n; i++) m[keys[i]].push_back(i);
  int k = 0;
  for (auto &kv : m) order_out[k++] = kv.first;
  return k;
}
void probe_sort(float *keys, int *payload, int n) {
  std::vector<std::pair<int, float>> v(n);
  for (int i = 0; i < n; i++) v[i] = {payload[i], keys[i]};
  std::sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.second < b.second; });
  for (int i = 0; i < n; i++) { payload[i] = v[i].first; keys[i] = v[i].second; }
}
void probe_uniform(unsigned seed, float lo, float hi, int n, float *out) {
  std::mt19937_64 rng; rng.seed(seed);
  for (int i = 0; i < n; i++) { std::uniform_real_distribution<float> d(lo, hi); out[i] = d(rng); }
}
void probe_rand(unsigned seed, int n, int *out) { std::srand(seed); for (int i = 0; i < n; i++) out[i] = std::rand(); }
}
