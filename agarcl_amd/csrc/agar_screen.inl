// Screen observation kernel: the reference's OpenGL frame, restated as rules.
//   camera        Renderer::camera_z / perspective_projection / view_projection (/root/reference/agario/rendering/renderer.hpp:91-120):
//                 eye above the player's mass-weighted centre at z = clamp(100 + mass/10, 100, 900), 45 degree vertical
//                 field of view => the visible half-height on the arena plane is z * tan(22.5 deg), half-width = that * W/H
//   frame         Renderer::render_screen (:163-185): white clear colour, then grid, pellets, foods, players in map order,
//                 viruses -- later draws overwrite earlier ones
//   shapes        triangle fans over regular polygons with vertices at angles k * 2 pi / N (core/renderables.hpp:198-207),
//                 N = 5 pellets, 7 foods, 50 cells, 150 viruses (core/Entities.hpp:13-16); a pixel belongs to a shape when its
//                 centre lies inside the polygon
//   grid          8 x 8 lines over the arena (renderer.hpp:26, renderables.hpp:245-339), colour (0.1, 0, 0) = byte 26, one pixel wide
//   read-back     glReadPixels(0, 0, W, H, GL_RGB, GL_UNSIGNED_BYTE) (rendering/FrameBufferObject.hpp:105): rows bottom-up,
//                 3 bytes per pixel; ScreenObservation exposes the same bytes as uint8 [1][W][H][3] (ScreenEnvironment.hpp:24-128)
// Colours: viruses are green (Entities.hpp:91), bots carry their class colour, cells take their player's colour.  The
// reference picks pellet / food / agent colours with rand() in the RENDERABLE build only (renderables.hpp:66, Player.hpp:53);
// the non-renderable engine this repo restates never makes those draws, so they are not reproducible: here the colour is
// palette[id % 6] (pellets, foods) and palette[pid % 6] (agents).  Parity is therefore rule-level (tolerance), and UNPINNED:
// no OpenGL context exists in the build container.
//
// agent_view: Renderer::multi_channel_render_screen (renderer.hpp:128-155) clears to (0,0,0,0) and draws by
// TYPE -- grid (0.1,0,0), pellets and foods (1,0,0), the main agent (0.9,0,0), other players (0,1,0), viruses (0,0,1),
// in that order -- reads RGBA back, and ScreenObservation::post_processing_frame_data (ScreenEnvironment.hpp:48-88)
// then walks the bytes once: values <= 230 (grid 26, main agent 230) move into the pixel's alpha byte, and a 255 pixel
// takes the previous pixel's alpha when the two previous pixels' alphas are both <= 30 (a sequential dependence along
// the flat buffer, row boundaries included).  "Main agent" = state.main_agent_pid = the last agent added (BaseEnvironment.hpp:189).
//
// One 256-thread workgroup per (arena, agent): wave 0 compacts the entities that can touch the view into an LDS list in
// draw order (ordered ballot compaction).  k_screen_obs then PAINTS them, in that order, into an LDS frame buffer: a band of rows at a
// time (<= 8192 pixels, the whole frame at 84 x 84), every wavefront owning a quarter of the band's rows and testing only the pixels
// of an entity's bounding box -- ~1200 inside-tests per 84 x 84 frame instead of 7056 pixels x ~40 entities -- the grid lines come
// from per-column / per-row flags, and the band leaves LDS as one coalesced byte stream.  Same per-pixel rules and the same fp32
// expressions as the pixel-wise kernel (k_screen_obs_pixelwise: every thread shades pixels, walking the whole list), which is kept as the
// cross-check (AGARCL_SCREEN_PIXELWISE=1; tests/test_screen_obs.py compares the two byte for byte).
#pragma once
#include "agar_types.h"

#define AG_SCR_CAP 512  // visible entities kept per frame (more are dropped from the END of the draw order)

struct AgScreenCfg { int W, H, agent_view;   // agent_view: the 4-channel frame of Renderer::multi_channel_render_screen
  int rows_per_wave, band_rows, fc_mul, fc_step;   // host arithmetic (scr_cfg_geometry): a wavefront's share of the rows, the rows of one LDS band -- three integer divisions less per wavefront
#ifdef AG_SCR_ABL   // measurement builds only (build.py --variant SCRABL -DAG_SCR_ABL): AGARCL_SCR_ABL=<bits> switches parts of k_screen_obs off
  int abl;          // 1 no entity list, 2 no painting, 4 no post-processing pass, 8 no global stores, 16 no background fill, 32 return at once, 64 / 128 return behind the first / second barrier
#endif
};
#ifdef AG_SCR_ABL
#define SCR_ABL(bit) (o.abl & (bit))
#else
#define SCR_ABL(bit) 0
#endif

#ifndef AGAR_CPU_EMU
// Renderer::camera_z: clamp(100 + mass / 10, 100, 900) in double, then float (renderer.hpp:91-99).  In fp32 without the double division: for
// mass < 8000 the float nearest to (1000 + mass) / 10 IS what the double expression rounds to -- (1000 + mass) / 10 is either a float itself or at
// least 1 / (10 * 2^18) away from the nearest midpoint between two floats of that size, eleven orders of magnitude more than the double
// expression's error -- and from 8000 on the clamp gives 900 (a wrapped mass of billions included)
__device__ __forceinline__ float scr_camera_z(unsigned mass) {
  if (mass >= 8000u) return 900.0f;
  const float z = (float)(1000u + mass) / 10.0f;
  return z < 100.0f ? 100.0f : z;
}
__device__ __forceinline__ unsigned scr_palette(int k) {  // core/color.hpp:4-12 as 0xBBGGRR bytes (GL rounds c * 255 to nearest)
  const unsigned pal[6] = {0x0000FFu /*red*/, 0x00A6FFu /*orange 1,.65,0*/, 0x00FFFFu /*yellow*/, 0x00FF00u /*green*/, 0xFF0000u /*blue*/, 0xCC3399u /*purple .6,.2,.8*/};
  return pal[((k % 6) + 6) % 6];
}
// Between the inscribed and the circumscribed circle the polygon's edge decides: the pixel is inside iff its projection on the outward normal of
// the edge of ITS sector -- at angle (k + 1/2) step -- is at most the apothem.  That edge is the one with the LARGEST projection, so for the two
// small polygons (pellets: 5 sides, foods: 7; a pentagon's ring between the circles is a third of its disc, nearly every pellet tile has a pixel
// in it) the test is the maximum over the edges with their normals as constants, mirrored in y: seven instructions instead of atan2f, cosf and
// sinf (~200, which every lane of the tile sat through: found in the ISA, r05).  Cells (50 sides) and viruses (150) keep the sector form.
__device__ __forceinline__ bool scr_edge_inside(float dx, float dy, float apo, int nsides) {
  if (nsides == 5) {   // normals at 36, 108, 180 degrees and their mirror images
    const float ay = fabsf(dy);
    const float p1 = dx * 0.8090169943749475f + ay * 0.5877852522924731f, p2 = dx * -0.30901699437494734f + ay * 0.9510565162951536f;
    return fmaxf(fmaxf(p1, p2), -dx) <= apo;
  }
  if (nsides == 7) {   // normals at 25.71, 77.14, 128.57, 180 degrees and their mirror images
    const float ay = fabsf(dy);
    const float p1 = dx * 0.9009688679024191f + ay * 0.4338837391175581f, p2 = dx * 0.2225209339563144f + ay * 0.9749279121818236f, p3 = dx * -0.6234898018587335f + ay * 0.7818314824680298f;
    return fmaxf(fmaxf(p1, p2), fmaxf(p3, -dx)) <= apo;
  }
  const float step = 6.28318530717958647692f / (float)nsides;
  float th = atan2f(dy, dx); if (th < 0.0f) th += 6.28318530717958647692f;
  float k = floorf(th / step);
  float phi = (k + 0.5f) * step;
  return dx * cosf(phi) + dy * sinf(phi) <= apo;
}
// cos(pi / nsides): the apothem of a unit polygon -- constants for the four polygons the renderer draws (core/Entities.hpp:13-16), cosf otherwise
__device__ __forceinline__ float scr_cos_half_step(int nsides) {
  if (nsides == 5) return 0.8090169943749475f;
  if (nsides == 7) return 0.9009688679024191f;
  if (nsides == 50) return 0.9980267284282716f;
  if (nsides == 150) return 0.9997806834748455f;
  return cosf(0.5f * (6.28318530717958647692f / (float)nsides));
}
__device__ __forceinline__ bool scr_inside(float dx, float dy, float r, int nsides) {
  float d2 = dx * dx + dy * dy;
  if (d2 > r * r) return false;
  float apo = r * scr_cos_half_step(nsides);
  if (d2 <= apo * apo) return true;
  return scr_edge_inside(dx, dy, apo, nsides);
}

// the same test with the apothem r * cos(pi / nsides) handed in (computed once per entity by the band kernel)
__device__ __forceinline__ bool scr_inside_apo(float dx, float dy, float r, float apo, int nsides) {
  float d2 = dx * dx + dy * dy;
  if (d2 > r * r) return false;
  if (d2 <= apo * apo) return true;
  return scr_edge_inside(dx, dy, apo, nsides);
}

__global__ void __launch_bounds__(256) k_screen_obs_pixelwise(const AgState *__restrict__ gs, AgScreenCfg o, uint8_t *out) {
  __shared__ float ex[AG_SCR_CAP], ey[AG_SCR_CAP], er[AG_SCR_CAP];
  __shared__ unsigned ec[AG_SCR_CAP];  // 0x00BBGGRR | nsides << 24
  __shared__ int n_list;
  const int na = gs->d.n_agents, arena = (int)blockIdx.x / na, agent = (int)blockIdx.x % na, P = gs->d.P;
  const int CH = o.agent_view ? 4 : 3;
  uint8_t *dst = out + (size_t)blockIdx.x * o.W * o.H * CH;
  float px, py; unsigned mass;
  obs_player(gs, arena, agent, px, py, mass);
  const float z = scr_camera_z(mass), half_h = z * 0.41421356237309504880f, half_w = half_h * ((float)o.W / (float)o.H);
  const float Wd = gs->g.W;
  const int ag_ts_lg = gs->d.ts_lg;
  const int32_t *ar = AG_AR_PTR(gs, arena);
  if (threadIdx.x < 64) {  // ---- wave 0: visible entities in draw order ----
    const int lane = (int)threadIdx.x; const unsigned long long lt = (1ull << lane) - 1ull;
    int count = 0;
    auto emit = [&](bool valid, float x, float y, float r, unsigned col) {
      bool vis = valid && fabsf(x - px) <= half_w + r && fabsf(y - py) <= half_h + r;
      unsigned long long m = __ballot(vis);
      int slot = count + __popcll(m & lt);
      if (vis && slot < AG_SCR_CAP) { ex[slot] = x; ey[slot] = y; er[slot] = r; ec[slot] = col; }
      count += __popcll(m);
    };
    const float *pxy = gs->pel_xy + (size_t)arena * gs->d.PC * 2; const int32_t *pid = gs->pel_id + (size_t)arena * gs->d.PC;
    const int np = ar[AG_TW(AR_NPEL)], nf = ar[AG_TW(AR_NFOOD)], nv = ar[AG_TW(AR_NVIR)];
    const float r_pel = gs->lut_r[AG_PELLET_MASS], r_food = gs->lut_r[AG_FOOD_MASS];
    const bool av = o.agent_view != 0;
    for (int b = 0; b < np; b += 64) { int i = b + lane; bool v = i < np; emit(v, v ? pxy[2 * i] : 0.f, v ? pxy[2 * i + 1] : 0.f, r_pel, v ? ((av ? 0x0000FFu : scr_palette(pid[i])) | (5u << 24)) : 0u); }
    { size_t fo = (size_t)arena * gs->d.FC;
      for (int b = 0; b < nf; b += 64) { int i = b + lane; bool v = i < nf; emit(v, v ? gs->food_x[fo + i] : 0.f, v ? gs->food_y[fo + i] : 0.f, r_food, v ? ((av ? 0x0000FFu : scr_palette(gs->food_id[fo + i])) | (7u << 24)) : 0u); } }
    const int main_slot = na - 1;  // state.main_agent_pid: the last agent added
    for (int kk = av ? -1 : 0; kk < P; kk++) {  // players in the engine's iteration order (agent view: the main agent first), cells in vector order
      const int slot = kk < 0 ? main_slot : ar[AG_TW(AR_ORDER0 + kk)];
      if (av && kk >= 0 && slot == main_slot) continue;
      const int32_t *pl = AG_PL_PTR(gs, arena, slot);
      const uint32_t *C = AG_CELLS_PTR(gs, arena, slot);
      const int n = pl[AG_TW(PL_NCELLS)], kind = pl[AG_TW(PL_KIND)];
      const unsigned col = (av ? (kk < 0 ? 0x0000E6u /* 0.9 -> 230 */ : 0x00FF00u)
                               : (kind == 0 ? scr_palette(pl[AG_TW(PL_PID)]) : kind == 1 ? scr_palette(4) : kind == 2 ? scr_palette(5) : kind == 3 ? scr_palette(0) : scr_palette(1))) | (50u << 24);
      bool v = lane < n;
      unsigned m = v ? C[AG_CELL_W(CF_M, lane)] : 0u;
      emit(v, v ? __uint_as_float(C[AG_CELL_W(CF_X, lane)]) : 0.f, v ? __uint_as_float(C[AG_CELL_W(CF_Y, lane)]) : 0.f, v ? gs->lut_r[m < AG_LUT_SIZE ? m : AG_LUT_SIZE - 1] : 0.f, col);
    }
    { size_t vo = (size_t)arena * gs->d.VC;
      for (int b = 0; b < nv; b += 64) { int i = b + lane; bool v = i < nv; unsigned m = v ? (unsigned)gs->vir_mass[vo + i] : 0u;
        emit(v, v ? gs->vir_x[vo + i] : 0.f, v ? gs->vir_y[vo + i] : 0.f, v ? gs->lut_r[m < AG_LUT_SIZE ? m : AG_LUT_SIZE - 1] : 0.f, (av ? 0xFF0000u : scr_palette(3)) | (150u << 24)); } }
    if (lane == 0) n_list = count < AG_SCR_CAP ? count : AG_SCR_CAP;
  }
  __syncthreads();
  const int n = n_list, NPIX = o.W * o.H;
  // grid lines: the pixel column / row a line falls into (one pixel wide)
  const float sx_scale = (float)o.W * 0.5f / half_w, sy_scale = (float)o.H * 0.5f / half_h, spacing = Wd / 7.0f;
  for (int pix = (int)threadIdx.x; pix < NPIX; pix += 256) {
    const int row = pix / o.W, col = pix - row * o.W;  // row 0 = bottom (glReadPixels)
    const float wx = px + (((float)col + 0.5f) / (float)o.W * 2.0f - 1.0f) * half_w;
    const float wy = py + (((float)row + 0.5f) / (float)o.H * 2.0f - 1.0f) * half_h;
    unsigned c = o.agent_view ? 0u : 0xFFFFFFu; bool drawn = false;
    const bool in_x = wx >= 0.0f && wx <= Wd, in_y = wy >= 0.0f && wy <= Wd;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float g = (float)i * spacing;
      int gc = (int)floorf((g - px) * sx_scale + (float)o.W * 0.5f), gr = (int)floorf((g - py) * sy_scale + (float)o.H * 0.5f);
      if ((gc == col && in_y) || (gr == row && in_x)) { c = 0x00001Au; drawn = true; }  // (0.1, 0, 0) -> 26
    }
    for (int k = 0; k < n; k++) {
      unsigned e = ec[k];
      if (scr_inside(wx - ex[k], wy - ey[k], er[k], (int)(e >> 24))) { c = e & 0xFFFFFFu; drawn = true; }
    }
    dst[(size_t)pix * CH] = (uint8_t)(c & 0xFF); dst[(size_t)pix * CH + 1] = (uint8_t)((c >> 8) & 0xFF); dst[(size_t)pix * CH + 2] = (uint8_t)((c >> 16) & 0xFF);
    if (CH == 4) dst[(size_t)pix * 4 + 3] = drawn ? 255 : 0;  // fragments are written with alpha 1
  }
  if (CH == 4) {  // ScreenObservation::post_processing_frame_data, a strictly sequential pass: one thread
    __syncthreads();
    if (threadIdx.x == 0) {
      int a1 = 255, a2 = 255;  // final alphas of the two previous pixels (none yet: prev_prev_Gline_index < 0 fails the test)
      for (int p = 0; p < NPIX; p++) {
        uint8_t *q = dst + (size_t)p * 4;
        int alpha = q[3];
        for (int ch = 0; ch < 3; ch++) {
          const int v = q[ch];
          if (v == 0) continue;
          if (v <= 230) { alpha = v; q[ch] = 0; }
          else if (p >= 2 && a2 <= 30 && a1 <= 30) alpha = a1;
        }
        q[3] = (uint8_t)alpha;
        a2 = a1; a1 = alpha;
      }
    }
  }
}
// ---- round 6: every wavefront rasterises its own rows -----------------------------------------------------------------------------------------
// Round 5's kernel walked a frame band by band, the four wavefronts sharing each band (four workgroup barriers per band); its ablation
// (scripts/gpu_screen_ablate.py, 4096 frames of 128 x 128 x 4 on a task-3 state, 147 us) said where the time went: 24 us before the first pixel (a chain of dependent global loads: cell
// count, then cells; arena words, then lists), 72 us painting ~35 entities (one dependent LDS round trip after another per entity, three wavefronts
// waiting at the barrier for the one that has the entities), 50 us in the agent view's run pass, 38 us filling and storing -- latencies in series, not
// instructions.  Here
//   * everything a frame reads is requested at once (capacities, not counts, bound the loads), the camera centre is summed from registers;
//   * after the entity list is complete a wavefront never waits for another: it owns H/4 consecutive rows and a private LDS band, fills, paints,
//     post-processes and stores them band by band at its own pace -- wavefronts of a workgroup and of the workgroups sharing a SIMD drift apart and
//     fill each other's latencies;
//   * the painter takes 64 entities' parameters in one LDS round trip (a lane each) and hands them out with v_readlane;
//   * the agent view's sequential pass needs the final alphas of the two pixels in front of a wavefront's first row: it paints those two pixels
//     itself; only when one of them is a 255-pixel (its alpha then depends on the run it belongs to) does it wait for the wavefront above to publish
//     them.
// Same per-pixel rules and fp32 expressions as before: byte-identical to k_screen_obs_pixelwise (tests/test_screen_obs.py).
// A frame is written once and read by somebody else, later: non-temporal stores (they do not allocate in the L2 the engine's state lives in).  Measured on 4096
// frames of 128 x 128 x 4, task-3 / 1 / 6 states: 114 / 132 / 158 -> 106 / 125 / 154 us
#ifdef AG_SCR_PLAIN_STORES
#define AG_SCR_STORE(p, v) (*(p) = (v))
#else
#define AG_SCR_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif
#ifndef AG_SCR_WAVES
#define AG_SCR_WAVES 5   // wavefronts per SIMD = workgroups per compute unit the small-frame instantiation is built for
#endif
#ifndef AG_SCR_WBAND
#define AG_SCR_WBAND 1024  // pixels of one wavefront's LDS band (4 KiB of packed RGBA): 8 rows of 128 -- a wavefront's 32 rows are four bands --, 12 rows of 84 (a quarter of
                           // 84 rows = 21 = two bands).  31.4 KB of LDS per workgroup: five per compute unit, what the 96 registers admit.  Every band costs a round of the
                           // painter's list reads, a fill, a reduction of the marks, a run pass and a store loop: 924 pixels (7 rows of 128, five bands) measured
                           // 121 / 142 / 174 us per 4096 frames of 128 x 128 x 4 on task-3 / 1 / 6 states against 113 / 134 / 160
#endif
// the band geometry of a frame (host side, at launch): rows per wavefront, rows per band (bands of equal size: ceil(rows / ceil(rows / rows that fit)))
static inline void scr_cfg_geometry(AgScreenCfg &o) {
  const int WB = (o.W > 256 || o.H > 256) ? 1024 : AG_SCR_WBAND;
  const int rpw = (o.H + 3) >> 2, max_rows = WB / o.W > 0 ? WB / o.W : 1, nbands = (rpw + max_rows - 1) / max_rows;
  o.rows_per_wave = rpw; o.band_rows = (rpw + nbands - 1) / nbands;
  const int gpr = o.W >> 2 > 0 ? o.W >> 2 : 1;            // the fill's groups of four pixels per row: lane / gpr as (lane * fc_mul) >> 16 (exact for lane < 64, gpr <= 64)
  o.fc_mul = (65536 + gpr - 1) / gpr; o.fc_step = 64 / gpr;
}
// Pixel box of an entity, first | last << 16; empty: first > last.  Pixel column c's centre lies at px + ((c + 1/2) / W * 2 - 1) half_w, so in the coordinate
// u = (x - px) kx + W/2 - 1/2, kx = W / (2 half_w), column c sits at u = c and the disc spans u -+ r kx: the columns ceil(low end) .. floor(high end), taken with
// AG_SCR_BOX_EPS of a pixel either side -- the inside test decides (in world coordinates, from colx[] / rowy[]), the box only has to contain what it accepts,
// and the two formulations differ by rounding: < 2e-3 pixel even for 1024-pixel frames of a 1000-unit arena.  A tight box is what makes small entities cheap: a
// pellet seen by a mass-1000 agent has a radius of 0.43 pixels -- it covers one pixel centre or none (none: the entity is never looked at again), where the
// box with a whole pixel of margin had nine candidates in three rows, each row a chunk for the agent view's run pass.
#ifndef AG_SCR_BOX_EPS
#define AG_SCR_BOX_EPS 0.03125f
#endif
__device__ __forceinline__ void scr_box(float x, float y, float r, float px, float py, float kx, float ky, int W, int H, unsigned &bx, unsigned &by) {
  const float ox = (float)W * 0.5f - 0.5f, oy = (float)H * 0.5f - 0.5f, dx = x - px, dy = y - py;
  int c0 = (int)ceilf((dx - r) * kx + ox - AG_SCR_BOX_EPS), c1 = (int)floorf((dx + r) * kx + ox + AG_SCR_BOX_EPS);
  int r0 = (int)ceilf((dy - r) * ky + oy - AG_SCR_BOX_EPS), r1 = (int)floorf((dy + r) * ky + oy + AG_SCR_BOX_EPS);
  c0 = c0 < 0 ? 0 : c0; c1 = c1 > W - 1 ? W - 1 : c1; r0 = r0 < 0 ? 0 : r0; r1 = r1 > H - 1 ? H - 1 : r1;
  if (c0 > c1 || r0 > r1) { c0 = 1; c1 = 0; r0 = 1; r1 = 0; }
  bx = (unsigned)c0 | ((unsigned)c1 << 16); by = (unsigned)r0 | ((unsigned)r1 << 16);
}
// what a fragment of colour e (0xNNBBGGRR, NN = polygon sides) leaves in a packed pixel.  Agent view: a value <= 230 is written as
// post_processing_frame_data leaves it (alpha = the value, channel cleared); a 255-colour stays a 255-pixel with alpha 255 until the run pass
template <bool AGV> __device__ __forceinline__ unsigned scr_paint_word(unsigned e) {
  return (AGV && (e & 0xFFFFFFu) <= 230u) ? ((e & 0xFFu) << 24) : ((e & 0xFFFFFFu) | 0xFF000000u);
}
template <int TAB, bool AGV> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TAB > 256 ? 4 : AG_SCR_WAVES, TAB > 256 ? 4 : AG_SCR_WAVES))) k_screen_obs(const AgState S, AgScreenCfg o, uint8_t *out) {
  // (the env's descriptor travels BY VALUE: its fields are scalar loads from the kernel-argument segment, one round trip less in front of the first state load
  // than through the descriptor's copy in HBM -- every workgroup's prologue is a chain of dependent round trips)
  const AgState *const gs = &S;
  constexpr int WB = TAB > 256 ? 1024 : AG_SCR_WBAND;   // (a band holds at least one row)
  constexpr int CH = AGV ? 4 : 3;
  constexpr unsigned GRIDV = AGV ? 0x1A000000u : 0xFF00001Au, BACKV = AGV ? 0u : 0x00FFFFFFu;   // (0.1, 0, 0) -> 26; alpha byte: a fragment was written
  __shared__ float ex[AG_SCR_CAP], ey[AG_SCR_CAP], er[AG_SCR_CAP];
  __shared__ unsigned ec[AG_SCR_CAP];                        // 0x00BBGGRR | nsides << 24
  __shared__ unsigned ebx[AG_SCR_CAP], eby[AG_SCR_CAP];      // pixel box: first | last << 16 column / row
  __shared__ __align__(16) unsigned fbw[4][WB + 4];          // per wavefront: [2], [3] = the two pixels in front of the band (agent view), [4 ..] = the band
  __shared__ __align__(4) uint8_t colflag[TAB], rowflag[TAB];   // bit 0: a grid line falls into this pixel column / row; bit 1: the column / row lies inside the arena
  __shared__ float colx[TAB], rowy[TAB];                     // world coordinate of every pixel column's / row's centre
  __shared__ int wcnt[8];                                    // [0..3] pellets listed by wavefront w, [4] foods, [5] the whole list
  __shared__ unsigned pp_last[4][2]; __shared__ int pp_done[4];   // agent view: a wavefront's last two FINAL pixels, and that they are there
  if (SCR_ABL(32)) { if (threadIdx.x == 0) fbw[0][0] = 1u; return; }   // (measurement: the launch alone -- workgroups, LDS, registers -- without a load)
  const int na = gs->d.n_agents, arena = (int)blockIdx.x / na, agent = (int)blockIdx.x % na, P = gs->d.P;
  const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint8_t *dst = out + (size_t)blockIdx.x * o.W * o.H * CH;
  const int ag_ts_lg = gs->d.ts_lg;
  const int32_t *ar = AG_AR_PTR(gs, arena);
  // ---- everything the frame reads, requested together (slots behind a count hold valid memory: capacities bound the loads) ----
  const int PCp = gs->d.PC, FCp = gs->d.FC, VCp = gs->d.VC;
  const float *pxy = gs->pel_xy + (size_t)arena * PCp * 2; const int32_t *pid = gs->pel_id + (size_t)arena * PCp;
  const int np = ar[AG_TW(AR_NPEL)], nf = ar[AG_TW(AR_NFOOD)], nv = ar[AG_TW(AR_NVIR)];
  const int32_t *plA = AG_PL_PTR(gs, arena, agent); const uint32_t *CA = AG_CELLS_PTR(gs, arena, agent);
  const int n_own = plA[AG_TW(PL_NCELLS)];
  unsigned own_xu = 0u, own_yu = 0u, own_m = 0u;             // the observing agent's cell slots, a lane each (every wavefront: no exchange needed)
  if (lane < AG_CC) { own_xu = CA[AG_CELL_W(CF_X, lane)]; own_yu = CA[AG_CELL_W(CF_Y, lane)]; own_m = CA[AG_CELL_W(CF_M, lane)]; }
  // pellets: the 64-pellet chunks of the CAPACITY, a quarter per wavefront (<= 2048 pellets: <= 8 chunks each)
  const int nchk = PCp >> 6, per = (nchk + 3) >> 2, c_lo = wave * per;
  float pxs[8], pys[8]; int pids[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { pxs[j] = 0.f; pys[j] = 0.f; pids[j] = 0;
    if (j < per && c_lo + j < nchk) { const int i = (c_lo + j) * 64 + lane; pxs[j] = pxy[2 * i]; pys[j] = pxy[2 * i + 1]; if (!AGV) pids[j] = pid[i]; } }
  // foods (wavefront 1) and viruses (wavefront 0, which lists them behind the players): the first 128 / 64 slots in registers, the rest in the loops below
  float fx0 = 0.f, fy0 = 0.f, fx1 = 0.f, fy1 = 0.f; int fi0 = 0, fi1 = 0;
  const size_t fo = (size_t)arena * FCp, vo = (size_t)arena * VCp;
  if (wave == 1) { if (lane < FCp) { fx0 = gs->food_x[fo + lane]; fy0 = gs->food_y[fo + lane]; if (!AGV) fi0 = gs->food_id[fo + lane]; }
                   if (lane + 64 < FCp) { fx1 = gs->food_x[fo + lane + 64]; fy1 = gs->food_y[fo + lane + 64]; if (!AGV) fi1 = gs->food_id[fo + lane + 64]; } }
  float vx0 = 0.f, vy0 = 0.f; unsigned vm0 = 0u;
  // (wavefront 0 lists the players between the two barriers, where every round trip is on the workgroup's critical path: the iteration order and
  // every player's kind / pid / cell count are requested here, a lane per position / slot, and handed out with v_readlane in the loop)
  int pf_order = 0, pf_meta = 0;   // pf_meta = kind | cells << 4 | pid << 12
  if (wave == 0) {
    if (lane < VCp) { vx0 = gs->vir_x[vo + lane]; vy0 = gs->vir_y[vo + lane]; vm0 = (unsigned)gs->vir_mass[vo + lane]; }
    if (lane < P) { pf_order = ar[AG_TW(AR_ORDER0 + lane)]; const int32_t *plq = AG_PL_PTR(gs, arena, lane); pf_meta = (plq[AG_TW(PL_KIND)] & 15) | (plq[AG_TW(PL_NCELLS)] & 255) << 4 | plq[AG_TW(PL_PID)] << 12; }
  }
  const float r_pel = gs->lut_r[AG_PELLET_MASS], r_food = gs->lut_r[AG_FOOD_MASS];
  const float Wd = gs->g.W;
  // ---- camera: Player::x / y / mass (core/Player.hpp:102-126), sequential fp32 sums in cell order -- obs_player's arithmetic on the lanes' registers ----
  float px, py; unsigned mass;
  { float sx = 0.0f, sy = 0.0f; unsigned tm = 0;
    for (int i = 0; i < n_own; i++) {
      const float xi = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)own_xu, i)), yi = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)own_yu, i));
      const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)own_m, i); const float fm = (float)m;
      float t = xi * fm; sx += t; t = yi * fm; sy += t; tm += m;
    }
    px = sx / (float)tm; py = sy / (float)tm; mass = tm; }
  const float z = scr_camera_z(mass), half_h = z * 0.41421356237309504880f, half_w = half_h * ((float)o.W / (float)o.H);
  // (the masses have arrived: the radii of the agent's cells and of the first 64 viruses, second hop, requested now -- wavefront 0 uses them behind the barrier)
  float own_r = 0.f, vr0 = 0.f;
  if (wave == 0) { if (lane < AG_CC) own_r = gs->lut_r[own_m < AG_LUT_SIZE ? own_m : AG_LUT_SIZE - 1]; vr0 = gs->lut_r[vm0 < AG_LUT_SIZE ? vm0 : AG_LUT_SIZE - 1]; }
  // ---- which pixel columns / rows lie inside the arena, and their centres' world coordinates (the grid lines' bits follow behind the barrier) ----
  const float sx_scale = (float)o.W * 0.5f / half_w, sy_scale = (float)o.H * 0.5f / half_h, spacing = Wd / 7.0f;
  for (int k = tid; k < o.W; k += 256) { const float wx = px + (((float)k + 0.5f) / (float)o.W * 2.0f - 1.0f) * half_w; colflag[k] = (wx >= 0.0f && wx <= Wd) ? 2 : 0; colx[k] = wx; }
  for (int k = tid; k < o.H; k += 256) { const float wy = py + (((float)k + 0.5f) / (float)o.H * 2.0f - 1.0f) * half_h; rowflag[k] = (wy >= 0.0f && wy <= Wd) ? 2 : 0; rowy[k] = wy; }
  // ---- visible entities in draw order: pellets (all four wavefronts), foods (wavefront 1), players and viruses (wavefront 0) ----
  unsigned long long pvm[8]; int pcnt = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { pvm[j] = 0ull;
    if (j < per) { const int i = (c_lo + j) * 64 + lane; pvm[j] = __ballot(i < np && fabsf(pxs[j] - px) <= half_w + r_pel && fabsf(pys[j] - py) <= half_h + r_pel); pcnt += __popcll(pvm[j]); } }
  unsigned long long fvm0 = 0ull, fvm1 = 0ull;
  if (wave == 1) {
    fvm0 = __ballot(lane < nf && fabsf(fx0 - px) <= half_w + r_food && fabsf(fy0 - py) <= half_h + r_food);
    fvm1 = __ballot(lane + 64 < nf && fabsf(fx1 - px) <= half_w + r_food && fabsf(fy1 - py) <= half_h + r_food);
    int fc = __popcll(fvm0) + __popcll(fvm1);
    for (int b = 128; b < nf; b += 64) { const int i = b + lane; const bool v = i < nf; const float x = v ? gs->food_x[fo + i] : 0.f, y = v ? gs->food_y[fo + i] : 0.f;
      fc += __popcll(__ballot(v && fabsf(x - px) <= half_w + r_food && fabsf(y - py) <= half_h + r_food)); }
    if (lane == 0) wcnt[4] = fc;
  }
  if (lane == 0) { wcnt[wave] = pcnt; pp_done[wave] = 0; }
  if (SCR_ABL(1)) { pcnt = 0; }
  __syncthreads();
  if (SCR_ABL(64)) { if (threadIdx.x == 0) out[blockIdx.x] = (uint8_t)wcnt[0]; return; }   // (measurement: everything in front of the first barrier)
  auto put = [&](int slot, float x, float y, float r, unsigned col) {
    if (slot < AG_SCR_CAP) { ex[slot] = x; ey[slot] = y; er[slot] = r; ec[slot] = col; unsigned bx, by; scr_box(x, y, r, px, py, sx_scale, sy_scale, o.W, o.H, bx, by); ebx[slot] = bx; eby[slot] = by; }
  };
  const int n_pel = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3], n_food = wcnt[4];
  { int base = 0;
    for (int w = 0; w < wave; w++) base += wcnt[w];
#pragma unroll
    for (int j = 0; j < 8; j++) if (j < per && pvm[j]) {
      if ((pvm[j] >> lane) & 1ull) put(base + __popcll(pvm[j] & lt), pxs[j], pys[j], r_pel, (AGV ? 0x0000FFu : scr_palette(pids[j])) | (5u << 24));
      base += __popcll(pvm[j]);
    } }
  if (wave == 3 && lane < 16) {   // grid lines: the pixel column / row a line falls into (one pixel wide) -- eight columns, eight rows, a lane each
    const float g = (float)(lane & 7) * spacing;
    if (lane < 8) { const int gc = (int)floorf((g - px) * sx_scale + (float)o.W * 0.5f); if (gc >= 0 && gc < o.W) colflag[gc] |= 1; }
    else { const int gr = (int)floorf((g - py) * sy_scale + (float)o.H * 0.5f); if (gr >= 0 && gr < o.H) rowflag[gr] |= 1; }
  }
  if (wave == 1) {
    int base = n_pel;
    if ((fvm0 >> lane) & 1ull) put(base + __popcll(fvm0 & lt), fx0, fy0, r_food, (AGV ? 0x0000FFu : scr_palette(fi0)) | (7u << 24));
    base += __popcll(fvm0);
    if ((fvm1 >> lane) & 1ull) put(base + __popcll(fvm1 & lt), fx1, fy1, r_food, (AGV ? 0x0000FFu : scr_palette(fi1)) | (7u << 24));
    base += __popcll(fvm1);
    for (int b = 128; b < nf; b += 64) { const int i = b + lane; const bool v = i < nf; const float x = v ? gs->food_x[fo + i] : 0.f, y = v ? gs->food_y[fo + i] : 0.f;
      const unsigned long long m = __ballot(v && fabsf(x - px) <= half_w + r_food && fabsf(y - py) <= half_h + r_food);
      if ((m >> lane) & 1ull) put(base + __popcll(m & lt), x, y, r_food, (AGV ? 0x0000FFu : scr_palette(gs->food_id[fo + i])) | (7u << 24));
      base += __popcll(m); }
  }
  if (wave == 0) {
    int count = n_pel + n_food;
    auto emit = [&](bool valid, float x, float y, float r, unsigned col) {
      const bool vis = valid && fabsf(x - px) <= half_w + r && fabsf(y - py) <= half_h + r;
      const unsigned long long m = __ballot(vis);
      if (vis) put(count + __popcll(m & lt), x, y, r, col);
      count += __popcll(m);
    };
    const int main_slot = na - 1;  // state.main_agent_pid: the last agent added
    for (int kk = AGV ? -1 : 0; kk < P; kk++) {  // players in the engine's iteration order (agent view: the main agent first), cells in vector order
      const int slot = kk < 0 ? main_slot : __builtin_amdgcn_readlane(pf_order, kk);
      if (AGV && kk >= 0 && slot == main_slot) continue;
      const int meta = __builtin_amdgcn_readlane(pf_meta, slot), kind = meta & 15;
      const unsigned col = (AGV ? (kk < 0 ? 0x0000E6u /* 0.9 -> 230 */ : 0x00FF00u)
                                : (kind == 0 ? scr_palette(meta >> 12) : kind == 1 ? scr_palette(4) : kind == 2 ? scr_palette(5) : kind == 3 ? scr_palette(0) : scr_palette(1))) | (50u << 24);
      int n; unsigned xu, yu, m; float rad;
      if (slot == agent) { n = n_own; xu = own_xu; yu = own_yu; m = own_m; rad = own_r; }   // (already in registers)
      else { const uint32_t *C = AG_CELLS_PTR(gs, arena, slot); n = (meta >> 4) & 255; const bool v = lane < AG_CC;
             xu = v ? C[AG_CELL_W(CF_X, lane)] : 0u; yu = v ? C[AG_CELL_W(CF_Y, lane)] : 0u; m = v ? C[AG_CELL_W(CF_M, lane)] : 0u;
             rad = lane < n ? gs->lut_r[m < AG_LUT_SIZE ? m : AG_LUT_SIZE - 1] : 0.f; }
      const bool v = lane < n;
      emit(v, v ? __uint_as_float(xu) : 0.f, v ? __uint_as_float(yu) : 0.f, v ? rad : 0.f, col);
    }
    { const unsigned vcol = (AGV ? 0xFF0000u : scr_palette(3)) | (150u << 24);
      { const bool v = lane < nv; emit(v, vx0, vy0, v ? vr0 : 0.f, vcol); }
      for (int b = 64; b < nv; b += 64) { const int i = b + lane; const bool v = i < nv; const unsigned m = v ? (unsigned)gs->vir_mass[vo + i] : 0u;
        emit(v, v ? gs->vir_x[vo + i] : 0.f, v ? gs->vir_y[vo + i] : 0.f, v ? gs->lut_r[m < AG_LUT_SIZE ? m : AG_LUT_SIZE - 1] : 0.f, vcol); } }
    if (lane == 0) wcnt[5] = SCR_ABL(1) ? 0 : (count < AG_SCR_CAP ? count : AG_SCR_CAP);
  }
  __syncthreads();   // the list is complete: from here on no wavefront waits for another (but for the rare hand-over of the run pass)
  if (SCR_ABL(128)) { if (threadIdx.x == 0) out[blockIdx.x] = (uint8_t)wcnt[5]; return; }   // (measurement: ... and of the second)
  const int n = __builtin_amdgcn_readfirstlane(wcnt[5]);
  // ---- this wavefront's rows, band by band ----
  const int rpw = o.rows_per_wave, wr0 = wave * rpw, wr1 = (wr0 + rpw < o.H ? wr0 + rpw : o.H) - 1;   // row 0 = bottom (glReadPixels)
  if (wr0 > wr1) return;
  unsigned *fb = &fbw[wave][4];
  const int band_rows = o.band_rows;   // (scr_cfg_geometry)
  // one visible-entity batch: 64 entities' parameters, a lane each, in ONE LDS round trip; handed out with v_readlane
  // Agent view: the two pixels in front of this wavefront's first row, as painted (final unless a 255-pixel).  Only a 255-pixel among the first two of the
  // wavefront's rows ever asks for them (the run pass below): computed only when the entity list allows for one -- for most wavefronts it does not
  auto lookback = [&]() {
    unsigned lb0 = 0u, lb1 = 0u;
#pragma unroll 1
    for (int t = 0; t < 2; t++) {
      const int f = wr0 * o.W - 1 - t;                      // flat pixel index (t = 0: the previous pixel)
      unsigned w = 0xFF000000u;                             // no such pixel: alpha 255 fails the run rule exactly as its p >= 2 test does
      if (f >= 0) {
        const int rr = f / o.W, cc = f - rr * o.W;
        const uint8_t cf = colflag[cc], rf = rowflag[rr];
        w = (((cf & 1) && (rf & 2)) || ((rf & 1) && (cf & 2))) ? GRIDV : BACKV;
        const float wx = colx[cc], wy = rowy[rr];
        for (int k0 = 0; k0 < n; k0 += 64) {
          const int kk = k0 + lane; const bool in_ = kk < n; const int ki = in_ ? kk : 0;
          const unsigned e = ec[ki]; const float r = er[ki];
          const unsigned long long m = __ballot(in_ && scr_inside_apo(wx - ex[ki], wy - ey[ki], r, r * scr_cos_half_step((int)(e >> 24)), (int)(e >> 24)));
          if (m) w = scr_paint_word<AGV>((unsigned)__builtin_amdgcn_readlane((int)e, 63 - (int)__builtin_clzll(m)));   // the LAST draw that covers the pixel
        }
      }
      if (t == 0) lb0 = w; else lb1 = w;
    }
    if (((lb0 | lb1) & 0xFFFFFFu) != 0u && !SCR_ABL(4)) {   // a 255-pixel: its alpha is its run's -- the wavefront above knows (rare: an entity at the frame's right edge of exactly that row)
      while (__hip_atomic_load(&pp_done[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(8);
      lb0 = pp_last[wave - 1][0]; lb1 = pp_last[wave - 1][1];
    }
    if (lane == 0) { fb[-1] = lb0; fb[-2] = lb1; }
    ag_lds_order();
  };
  if (AGV) {   // can a 255-coloured entity cover one of this wavefront's first two pixels?  (its box holds row wr0 and column 0 or 1; frames one pixel wide: always)
    bool need = o.W < 2;
    for (int k0 = 0; k0 < n && !need; k0 += 64) {
      const int kk = k0 + lane; const bool in_ = kk < n; const int ki = in_ ? kk : 0;
      const unsigned bx = ebx[ki], by = eby[ki], e = ec[ki];
      need = __ballot(in_ && (e & 0xFFFFFFu) > 230u && (bx & 0xFFFFu) <= 1u && (bx & 0xFFFFu) <= (bx >> 16) && (int)(by & 0xFFFFu) <= wr0 && wr0 <= (int)(by >> 16)) != 0ull;
    }
    if (need) lookback();
  }
  // rows of up to 256 pixels, a multiple of four: a pass of the fill covers whole rows -- 64 / (W / 4) of them, a lane per four pixels, the lanes
  // beyond that idle (84 x 84: 63 of 64 work) -- so a lane's four columns never change.  Its masks: byte i = 1 where column i carries a grid line
  // (shown in rows inside the arena) / lies inside the arena (where a row's line shows)
  const int fc_gpr = o.W >> 2;
  const bool fixed_cols = (o.W & 3) == 0 && fc_gpr >= 1 && fc_gpr <= 64;
  unsigned fc_line = 0u, fc_inside = 0u; int fc_r = 0, fc_step = 1, fc_gc = 0; bool fc_on = false;
  if (fixed_cols) { fc_step = o.fc_step; fc_r = (lane * o.fc_mul) >> 16; fc_gc = lane - fc_r * fc_gpr; fc_on = fc_r < fc_step;
                    const unsigned cf4 = *(const unsigned *)&colflag[fc_gc << 2]; fc_line = cf4 & 0x01010101u; fc_inside = (cf4 & 0x02020202u) >> 1; }
  for (int row0 = wr0; row0 <= wr1; row0 += band_rows) {
    const int rows = wr1 + 1 - row0 < band_rows ? wr1 + 1 - row0 : band_rows, npix = rows * o.W, rlast = row0 + rows - 1;
    // background + grid: four pixels of a row per lane and store -- one row flag, the four column flags as one word, a 16-byte LDS store
    if (SCR_ABL(16)) {} else
    if (fixed_cols) {   // the lane keeps its column group from pass to pass: its two column masks were taken once, in front of the band loop
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      if (fc_on) for (int rb = fc_r; rb < rows; rb += fc_step) {
        const unsigned rf = rowflag[row0 + rb];
        const unsigned m = ((rf & 2u) ? fc_line : 0u) | ((rf & 1u) ? fc_inside : 0u);                    // byte i: pixel i of the group is a grid pixel
        v4u v; v.x = (m & 0x01u) ? GRIDV : BACKV; v.y = (m & 0x0100u) ? GRIDV : BACKV; v.z = (m & 0x010000u) ? GRIDV : BACKV; v.w = (m & 0x01000000u) ? GRIDV : BACKV;
        *(v4u *)&fb[(rb * fc_gpr + fc_gc) << 2] = v;
      }
    } else if ((o.W & 3) == 0) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const int gpr = o.W >> 2, ngrp = npix >> 2;   // groups per row, groups in the band
      const int step_r = 64 / gpr, step_c = 64 - step_r * gpr;
      for (int g = lane, r = lane / gpr, gc = lane - r * gpr; g < ngrp; g += 64, r += step_r, gc += step_c) {
        if (gc >= gpr) { gc -= gpr; r += 1; }
        const unsigned cf4 = *(const unsigned *)&colflag[gc << 2]; const unsigned rf = rowflag[row0 + r];
        const unsigned line = (rf & 2u) ? 0x01010101u : 0u, inside = (rf & 1u) ? 0x02020202u : 0u;     // a column's line shows in rows inside the arena; a row's line in columns inside
        const unsigned m = (cf4 & line) | ((cf4 & inside) >> 1);                                        // byte i: pixel i of the group is a grid pixel
        v4u v; v.x = (m & 0x01u) ? GRIDV : BACKV; v.y = (m & 0x0100u) ? GRIDV : BACKV; v.z = (m & 0x010000u) ? GRIDV : BACKV; v.w = (m & 0x01000000u) ? GRIDV : BACKV;
        *(v4u *)&fb[g << 2] = v;
      }
    } else {
      const int step_r = 64 / o.W, step_c = 64 - step_r * o.W;   // (one division: the loop walks rows and columns by addition)
      for (int q = lane, r = lane / o.W, cidx = lane - r * o.W; q < npix; q += 64, r += step_r, cidx += step_c) {
        while (cidx >= o.W) { cidx -= o.W; r += 1; }
        const uint8_t cf = colflag[cidx], rf = rowflag[row0 + r];
        fb[q] = (((cf & 1) && (rf & 2)) || ((rf & 1) && (cf & 2))) ? GRIDV : BACKV;
      }
    }
    ag_lds_order();
    // entities in draw order: later draws overwrite earlier ones (one wavefront, in order: no exchange)
    unsigned long long chunks255 = 0ull;   // agent view: the band's 64-pixel chunks that may hold a 255-pixel (bit c = chunk c; a band has <= 16)
    unsigned marks_acc = 0u;               // ... collected lane by lane, row by row (the chunks a box's row touches), OR-reduced once behind the painter
    if (!SCR_ABL(2)) for (int k0 = 0; k0 < n; k0 += 64) {
      const int kk = k0 + lane; const bool in_ = kk < n; const int ki = in_ ? kk : 0;
      const unsigned bx_ = in_ ? ebx[ki] : 1u, by_ = in_ ? eby[ki] : 1u;   // (1 = first 1, last 0: empty)
      const float x_ = ex[ki], y_ = ey[ki], r_ = er[ki]; const unsigned e_ = ec[ki];
#ifndef AG_SCR_LANE_MIN
#define AG_SCR_LANE_MIN 3
#endif
#ifndef AG_SCR_LANE_MIN_RGB
#define AG_SCR_LANE_MIN_RGB 6   // (the plain frame's form takes two passes of LDS atomics)
#endif
      constexpr int LANE_MIN = AGV ? AG_SCR_LANE_MIN : AG_SCR_LANE_MIN_RGB;   // hits of one run from which the lane-parallel form pays
      // this lane's entity's box inside the band
      unsigned tbx, tby;   // first | last << 16 again (only lanes with a non-empty box are ever read)
      unsigned long long hits;
      { const int c0 = (int)(bx_ & 0xFFFFu), c1 = (int)(bx_ >> 16); int r0 = (int)(by_ & 0xFFFFu), r1 = (int)(by_ >> 16);
        r0 = r0 < row0 ? row0 : r0; r1 = r1 > rlast ? rlast : r1;
        hits = __ballot(in_ && c0 <= c1 && r0 <= r1);
        tbx = bx_; tby = (unsigned)r0 | ((unsigned)r1 << 16); }
      while (hits) {
        // Many hits in one batch -- a line of pellets across the view (tasks 1 and 2 lay 350 of them along a square, one unit apart), a cloud of
        // ejected food, the pellets around a split agent: when their boxes are small every lane paints its own entity, <= 6 x 6 pixels, while the
        // others paint theirs (one wavefront-wide loop instead of one tile pass per entity).  Taken for the longest RUN of hits, from the first one on,
        // that are small and alike -- the draw order of the run against what comes before and behind it stays what it was.  Agent view: alike = all paint
        // the same word -- then the ORDER among them is immaterial, plain stores.  Plain frame: alike = the same polygon; colours differ by id, the latest
        // draw must win a pixel two entities cover -- the alpha byte, which a three-channel frame never outputs, carries the lane: a first pass clears it
        // on every covered pixel, a second pass takes the maximum of (lane + 1) << 24 | colour, which is the highest lane's, i.e. the latest entity's, word.
        const int j = (int)__builtin_ctzll(hits);
        const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)e_, j);
        unsigned long long run = 0ull;
        if (__popcll(hits) >= LANE_MIN) {
          const bool alike = AGV ? e_ == e0 : (e_ >> 24) == (e0 >> 24);   // (the polygon's side count is wave-uniform in the inside test)
          const unsigned long long bad = __ballot(((hits >> lane) & 1ull) && (!alike || (tbx >> 16) - (tbx & 0xFFFFu) > 5u || (tby >> 16) - (tby & 0xFFFFu) > 5u));
          run = bad ? hits & ((1ull << (int)__builtin_ctzll(bad)) - 1ull) : hits;
        }
        if (__popcll(run) >= LANE_MIN) {
          const bool mine = (run >> lane) & 1ull;
          unsigned bxv = tbx, byv = tby;
          asm volatile("" : "+v"(bxv), "+v"(byv));   // (opaque: keeps the unpacked box and everything derived from it inside this block instead of hoisted in front of the loop, where it cost 14 registers)
          const int c0 = (int)(bxv & 0xFFFFu), r0 = (int)(byv & 0xFFFFu), bw = (int)(bxv >> 16) - c0 + 1, bh = (int)(byv >> 16) - r0 + 1;
          {
          const int ns = (int)(e0 >> 24); const float apo = r_ * scr_cos_half_step(ns);
          float cx[6];
#pragma unroll
          for (int dx = 0; dx < 6; dx++) cx[dx] = colx[(mine && dx < bw) ? c0 + dx : 0] - x_;
          if (AGV) {
            const unsigned paint = scr_paint_word<AGV>(e0);
            const bool is255 = (e0 & 0xFFFFFFu) > 230u;
#pragma unroll
            for (int dy = 0; dy < 6; dy++) {
              if (__ballot(mine && dy < bh) == 0ull) break;
              const bool rowok = mine && dy < bh; const float yy = rowy[rowok ? r0 + dy : 0] - y_; const int rb = (r0 + dy - row0) * o.W + c0;
              if (is255 && rowok && bw > 0) { const int lo_ = rb >> 6, hi_ = (rb + bw - 1) >> 6; marks_acc |= ((2u << (hi_ - lo_)) - 1u) << lo_; }
#pragma unroll
              for (int dx = 0; dx < 6; dx++) if (rowok && dx < bw && scr_inside_apo(cx[dx], yy, r_, apo, ns)) fb[rb + dx] = paint;
            }
          } else {
            unsigned long long cover = 0ull;   // bit dy * 6 + dx: this lane's entity covers that pixel of its box
#pragma unroll
            for (int dy = 0; dy < 6; dy++) {
              if (__ballot(mine && dy < bh) == 0ull) break;
              const bool rowok = mine && dy < bh; const float yy = rowy[rowok ? r0 + dy : 0] - y_; const int rb = (r0 + dy - row0) * o.W + c0;
#pragma unroll
              for (int dx = 0; dx < 6; dx++) if (rowok && dx < bw && scr_inside_apo(cx[dx], yy, r_, apo, ns)) { cover |= 1ull << (dy * 6 + dx); atomicAnd(&fb[rb + dx], 0x00FFFFFFu); }
            }
            ag_lds_order();
            const unsigned word = ((unsigned)(lane + 1) << 24) | (e_ & 0xFFFFFFu);
#pragma unroll
            for (int dy = 0; dy < 6; dy++) {
              if (__ballot((cover >> (dy * 6)) & 63ull) == 0ull) continue;
              const int rb = (r0 + dy - row0) * o.W + c0;
#pragma unroll
              for (int dx = 0; dx < 6; dx++) if ((cover >> (dy * 6 + dx)) & 1ull) atomicMax(&fb[rb + dx], word);
            }
            ag_lds_order();
          }
          }
          hits &= ~run;
        } else {
        // one entity, the wavefront's lanes over the pixels of its (tight) box
        hits &= hits - 1ull;
        const unsigned bx = (unsigned)__builtin_amdgcn_readlane((int)tbx, j), by = (unsigned)__builtin_amdgcn_readlane((int)tby, j);
        const int c0 = (int)(bx & 0xFFFFu), c1 = (int)(bx >> 16), r0 = (int)(by & 0xFFFFu), r1 = (int)(by >> 16);
        const float x = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x_), j)), y = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(y_), j));
        const float r = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(r_), j)); const unsigned e = e0;
        const int ns = (int)(e >> 24); const float apo = r * scr_cos_half_step(ns);
        const unsigned paint = scr_paint_word<AGV>(e);
        if (AGV && (e & 0xFFFFFFu) > 230u) {
          if (r1 - r0 < 64) {   // a lane per row of the box: the chunks that row touches
            if (lane <= r1 - r0) { const int rb = (r0 + lane - row0) * o.W + c0, lo_ = rb >> 6, hi_ = (rb + c1 - c0) >> 6; marks_acc |= ((hi_ - lo_ >= 31) ? ~0u : ((2u << (hi_ - lo_)) - 1u)) << lo_; }
          } else {              // (bands of very narrow frames: every chunk from the box's first pixel to its last)
            const int lo = ((r0 - row0) * o.W + c0) >> 6, hi = ((r1 - row0) * o.W + c1) >> 6;
            chunks255 |= ((hi - lo >= 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull)) << lo;
          }
        }
        for (int ty = r0; ty <= r1; ty += 8) for (int tx = c0; tx <= c1; tx += 8) {   // 8 x 8 pixel tiles of the box, a lane per pixel
          const int rr = ty + (lane >> 3), cc = tx + (lane & 7);
          if (rr <= r1 && cc <= c1 && scr_inside_apo(colx[cc] - x, rowy[rr] - y, r, apo, ns)) fb[(rr - row0) * o.W + cc] = paint;
        }
        }
      }
    }
    ag_lds_order();
    if (AGV) chunks255 |= (unsigned long long)wred_or(marks_acc);
    if (AGV && !SCR_ABL(4)) {
      // ScreenObservation::post_processing_frame_data (ScreenEnvironment.hpp:48-88): a sequential pass over the flat buffer.  With the colours this kernel
      // paints (one non-zero channel per pixel: 26 grid, 230 main agent, 255 pellets / others / viruses) the pass has a closed form.  A pixel whose
      // channel is <= 230 moves it into alpha: no dependence -- the painter writes such pixels in that form at once.  A 255-pixel keeps alpha 255 unless
      // the two previous FINAL alphas are both <= 30, then it takes the previous one; so a run of consecutive 255-pixels takes ONE value, decided at its
      // first pixel -- 255, or the alpha in front of the run -- and only run starts are sequential.  The wavefront walks, in pixel order, the 64-pixel
      // chunks the painter marked; fb[-1] / fb[-2] hold the two final pixels in front of the band, so the first pixels look back like any other.
      int run_val = -1, prev_c = -2;   // value of the run that reaches into the current chunk from the left (-1: none); the chunk visited before this one
      for (unsigned long long cm = chunks255; cm; cm &= cm - 1ull) {
        const int cch = (int)__builtin_ctzll(cm), c0_ = cch << 6;
        if (cch != prev_c + 1) run_val = -1;                     // (the chunks in between hold no 255-pixel: no run crosses them)
        prev_c = cch;
        const int q = c0_ + lane; const bool in = q < npix; const int qi = in ? q : 0;
        // the chunk's words and, for the lanes that turn out to start a run, the two pixels in front: three reads in one round trip, no branches
        const unsigned w = fb[qi], wl1 = fb[qi - 1], wl2 = fb[qi - 2];
        const bool X = in && (w & 0xFFFFFFu) != 0u;            // only 255-pixels carry a colour
        const unsigned long long xm = __ballot(X);
        if (xm == 0ull) { run_val = -1; continue; }
        // All runs of the chunk at once: a 255-pixel finds the start of its run in the ballot (the highest non-255 pixel below it), the start decides
        // the run's value from the two pixels in front of it -- final already unless the second one is a 255-pixel of THIS chunk (two runs one pixel
        // apart: the chunk then takes the run-by-run loop below) -- and hands it to the run with one ds_bpermute.
        if (!((xm & ~(xm << 1)) & (xm << 2))) {   // (no two runs one pixel apart -- X[i], !X[i-1], X[i-2] --: scalar arithmetic on the ballot)
          const unsigned long long below = ~xm & lt;
          const int sp = below ? 64 - (int)__builtin_clzll(below) : 0;            // chunk position at which this lane's run starts (0: it reaches the left edge)
          const bool from_left = !below && run_val >= 0;                           // ... and continues the run of the chunk before
          const bool start = X && sp == lane && !from_left;
          const int f1 = (int)(wl1 >> 24), f2 = (int)(wl2 >> 24);
          const int val = (start && f2 <= 30 && f1 <= 30) ? f1 : 255;
          int mine = __builtin_amdgcn_ds_bpermute(sp << 2, val);
          if (from_left) mine = run_val;
          if (X) fb[q] = (w & 0xFFFFFFu) | ((unsigned)mine << 24);
          ag_lds_order();
          run_val = (xm >> 63) ? __builtin_amdgcn_readlane(mine, 63) : -1;
          continue;
        }
        unsigned long long todo = xm;
        int carry_val = run_val;
        while (todo) {
          const int s_ = (int)__builtin_ctzll(todo);           // first undecided 255-pixel: a run start, or the continuation of the left run
          const unsigned long long from = xm >> s_; const int len = (~from) ? (int)__builtin_ctzll(~from) : 64 - s_;   // length of the run inside this chunk
          int val;
          if (s_ == 0 && carry_val >= 0) val = carry_val;     // the run started in an earlier chunk
          else {
            const int f1 = (int)(fb[c0_ + s_ - 1] >> 24), f2 = (int)(fb[c0_ + s_ - 2] >> 24);   // final alphas of the two pixels in front of the run start
            val = (f2 <= 30 && f1 <= 30) ? f1 : 255;
          }
          if (lane >= s_ && lane < s_ + len) fb[q] = (w & 0xFFFFFFu) | ((unsigned)val << 24);
          ag_lds_order();
          const unsigned long long runmask = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << s_;
          todo &= ~runmask;
          carry_val = -1;
          run_val = (s_ + len == 64) ? val : -1;               // the run touches the chunk's right edge: it may continue
        }
        if (!(xm >> 63)) run_val = -1;
      }
      // the band's last two final pixels: in front of the next band, or published for the wavefront below
      { const unsigned l1 = fb[npix - 1], l2 = fb[npix - 2];   // (npix == 1: fb[-1], the pixel in front of this band -- which is what lies two before the next)
        ag_lds_order();
        if (lane == 0) {
          if (row0 + rows <= wr1) { fb[-1] = l1; fb[-2] = l2; }
          else { pp_last[wave][0] = l1; pp_last[wave][1] = l2; __hip_atomic_store(&pp_done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
        ag_lds_order(); }
    }
    // the band leaves LDS as one coalesced byte stream: whole 32-bit words (a packed RGBA pixel IS its four output bytes, 0xAABBGGRR little endian;
    // three-channel frames pack four pixels into three words); frames and bands start on 4- / 16-byte boundaries when their byte counts say so,
    // anything else takes the byte loop
    uint8_t *bd = dst + (size_t)row0 * o.W * CH; const int nbytes = npix * CH;
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    if (SCR_ABL(8)) {} else
    if (CH == 4 && (((size_t)bd) & 15) == 0 && (npix & 3) == 0) { v4u_ *bw = (v4u_ *)bd; const v4u_ *fw = (const v4u_ *)fb; for (int g = lane; g < (npix >> 2); g += 64) AG_SCR_STORE(&bw[g], fw[g]); }
    else if (CH == 3 && (((size_t)bd) & 3) == 0 && (npix & 3) == 0) {
      unsigned *bw = (unsigned *)bd; const v4u_ *fw = (const v4u_ *)fb;
      for (int g = lane; g < (npix >> 2); g += 64) {
        const v4u_ v = fw[g]; const unsigned p0 = v.x & 0xFFFFFFu, p1 = v.y & 0xFFFFFFu, p2 = v.z & 0xFFFFFFu, p3 = v.w & 0xFFFFFFu;
        AG_SCR_STORE(&bw[3 * g], p0 | (p1 << 24)); AG_SCR_STORE(&bw[3 * g + 1], (p1 >> 8) | (p2 << 16)); AG_SCR_STORE(&bw[3 * g + 2], (p2 >> 16) | (p3 << 8));
      }
    }
    else if (CH == 4 && (((size_t)bd) & 3) == 0) { unsigned *bw = (unsigned *)bd; for (int q = lane; q < npix; q += 64) bw[q] = fb[q]; }
    else if (CH == 3 && (((size_t)bd) & 3) == 0) {
      unsigned *bw = (unsigned *)bd; const int nwords = nbytes >> 2;
      for (int j = lane; j < nwords; j += 64) {
        // output bytes 4j .. 4j+3 = channels (4j + t) % 3 of pixels (4j + t) / 3: 4j = 3 q0 + c0  (c0 <= 2: the word's last byte is byte 5 of the two pixels' six)
        const int b0 = 4 * j, q0 = b0 / 3, c0 = b0 - 3 * q0;
        const unsigned long long two = (unsigned long long)(fb[q0] & 0xFFFFFFu) | ((unsigned long long)(fb[q0 + 1 < npix ? q0 + 1 : q0] & 0xFFFFFFu) << 24);
        bw[j] = (unsigned)(two >> (8 * c0));
      }
      for (int b = 4 * nwords + lane; b < nbytes; b += 64) { const int q = b / 3, ch = b - q * 3; bd[b] = (uint8_t)((fb[q] >> (8 * ch)) & 0xFFu); }
    }
    else if (CH == 4) { for (int b = lane; b < nbytes; b += 64) bd[b] = (uint8_t)((fb[b >> 2] >> (8 * (b & 3))) & 0xFFu); }
    else { for (int b = lane; b < nbytes; b += 64) { const int q = b / 3, ch = b - q * 3; bd[b] = (uint8_t)((fb[q] >> (8 * ch)) & 0xFFu); } }
    ag_lds_order();   // (the next band's fill comes after these reads)
  }
}
#endif
