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
#ifdef AG_SCR_ABL   // measurement builds only (build.py --variant SCRABL -DAG_SCR_ABL): AGARCL_SCR_ABL=<bits> switches parts of k_screen_obs off
  int abl;          // 1 no entity list, 2 no painting, 4 no post-processing pass, 8 no global stores, 16 no background fill
#endif
};
#ifdef AG_SCR_ABL
#define SCR_ABL(bit) (o.abl & (bit))
#else
#define SCR_ABL(bit) 0
#endif

#ifndef AGAR_CPU_EMU
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
  double zd = 100.0 + (double)mass / 10.0; zd = zd < 100.0 ? 100.0 : (zd > 900.0 ? 900.0 : zd);
  const float z = (float)zd, half_h = z * 0.41421356237309504880f, half_w = half_h * ((float)o.W / (float)o.H);
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
// pixels of one LDS band (14 KiB of packed RGBA): half an 84 x 84 frame.  With the whole frame in one band (7168 pixels, 28 KiB) the kernel
// holds 52 KB of LDS and three workgroups share a compute unit; with two bands of 42 rows it holds 38 KB and four do -- the second pass over
// the entity boxes costs less than the fourth workgroup brings: 4096 frames 215.8 -> 196.4 us, agent view 285.8 -> 254.8 (four bands of 21
// rows, five workgroups: 214.2 / 269.9)
#ifndef AG_SCR_BAND
#define AG_SCR_BAND 3584
#endif
// TAB: capacity of the per-column / per-row tables (256 for frames up to 256 x 256 -- with it the kernel holds 31 KB of LDS and FIVE workgroups
// share a compute unit --, 1024 beyond)
// AGV: the 4-channel agent-view frame (a template parameter since r05: the plain frame carries none of its tests)
template <int TAB, bool AGV> __global__ void __launch_bounds__(256) k_screen_obs(const AgState *__restrict__ gs, AgScreenCfg o, uint8_t *out) {
  __shared__ float ex[AG_SCR_CAP], ey[AG_SCR_CAP], er[AG_SCR_CAP];
  __shared__ unsigned ec[AG_SCR_CAP];  // 0x00BBGGRR | nsides << 24
  __shared__ __align__(16) unsigned fb[AG_SCR_BAND];  // 0xAABBGGRR of the band's pixels
  __shared__ __align__(4) uint8_t colflag[TAB], rowflag[TAB];   // bit 0: a grid line falls into this pixel column / row; bit 1: the column / row lies inside the arena
  __shared__ float colx[TAB], rowy[TAB];           // world coordinate of every pixel column's / row's centre
  __shared__ float eapo[AG_SCR_CAP];                 // apothem of an entity's polygon
  __shared__ unsigned ebx[AG_SCR_CAP], eby[AG_SCR_CAP];   // pixel box of an entity: first | last << 16 column / row (one pixel of margin; empty: first > last)
  __shared__ int n_list;
  __shared__ unsigned long long pp_chunks;   // agent view: the band's 64-pixel chunks that may hold a 255-pixel (bit c = chunk c; a band has <= 56)
  const int na = gs->d.n_agents, arena = (int)blockIdx.x / na, agent = (int)blockIdx.x % na, P = gs->d.P;
  constexpr int CH = AGV ? 4 : 3;
  uint8_t *dst = out + (size_t)blockIdx.x * o.W * o.H * CH;
  float px, py; unsigned mass;
  obs_player(gs, arena, agent, px, py, mass);
  double zd = 100.0 + (double)mass / 10.0; zd = zd < 100.0 ? 100.0 : (zd > 900.0 ? 900.0 : zd);
  const float z = (float)zd, half_h = z * 0.41421356237309504880f, half_w = half_h * ((float)o.W / (float)o.H);
  const float Wd = gs->g.W;
  const int ag_ts_lg = gs->d.ts_lg;
  const int32_t *ar = AG_AR_PTR(gs, arena);
  // ---- visible entities in draw order ----
  // The pellets -- most of the list -- are compacted by all four wavefronts (r05: the kernel is bound by what a SIMD issues, a workgroup's
  // wavefronts sit on different SIMDs, and work that only wavefront 0 does is issued by one SIMD in four): each takes a quarter of the 64-pellet
  // chunks, counts what it will list, and writes behind the wavefronts in front of it.  Foods, cells and viruses follow on wavefront 0.
  __shared__ int wcnt[4];
  const float *pxy = gs->pel_xy + (size_t)arena * gs->d.PC * 2; const int32_t *pid = gs->pel_id + (size_t)arena * gs->d.PC;
  const int np = ar[AG_TW(AR_NPEL)];
  const float r_pel = gs->lut_r[AG_PELLET_MASS];
  constexpr bool av = AGV;
  int listed_pellets;
  {
    // (the wavefront's number is wave-uniform: said so, or the compiler keeps everything derived from it in vector registers and walks the loops with exec masks)
    const int lw = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), ll = (int)threadIdx.x & 63; const unsigned long long llt = (1ull << ll) - 1ull;
    const int nchk = (np + 63) >> 6, per = (nchk + 3) >> 2, c_lo = lw * per, c_hi = (lw + 1) * per < nchk ? (lw + 1) * per : nchk;   // (<= 2048 pellets: per <= 8)
    float xs[8], ys[8]; int ids[8]; unsigned long long vm[8]; int cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { xs[j] = 0.f; ys[j] = 0.f; ids[j] = 0; vm[j] = 0ull;
      if (j < per) { const int i = (c_lo + j) * 64 + ll; const bool v = c_lo + j < c_hi && i < np; if (v) { xs[j] = pxy[2 * i]; ys[j] = pxy[2 * i + 1]; if (!av) ids[j] = pid[i]; } } }
#pragma unroll
    for (int j = 0; j < 8; j++) if (j < per) { const int i = (c_lo + j) * 64 + ll; const bool v = c_lo + j < c_hi && i < np;
      vm[j] = __ballot(v && fabsf(xs[j] - px) <= half_w + r_pel && fabsf(ys[j] - py) <= half_h + r_pel); cnt += __popcll(vm[j]); }
    if (ll == 0) wcnt[lw] = cnt;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < lw; w++) base += wcnt[w];
    listed_pellets = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
#pragma unroll
    for (int j = 0; j < 8; j++) if (j < per && vm[j]) {
      const int slot = base + __popcll(vm[j] & llt);
      if (((vm[j] >> ll) & 1ull) && slot < AG_SCR_CAP) { ex[slot] = xs[j]; ey[slot] = ys[j]; er[slot] = r_pel; ec[slot] = (av ? 0x0000FFu : scr_palette(ids[j])) | (5u << 24); }
      base += __popcll(vm[j]);
    }
  }
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x; const unsigned long long lt = (1ull << lane) - 1ull;
    int count = listed_pellets;
    auto emit = [&](bool valid, float x, float y, float r, unsigned col) {
      bool vis = valid && fabsf(x - px) <= half_w + r && fabsf(y - py) <= half_h + r;
      unsigned long long m = __ballot(vis);
      int slot = count + __popcll(m & lt);
      if (vis && slot < AG_SCR_CAP) { ex[slot] = x; ey[slot] = y; er[slot] = r; ec[slot] = col; }
      count += __popcll(m);
    };
    const int nf = ar[AG_TW(AR_NFOOD)], nv = ar[AG_TW(AR_NVIR)];
    const float r_food = gs->lut_r[AG_FOOD_MASS];
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
    if (lane == 0) n_list = SCR_ABL(1) ? 0 : (count < AG_SCR_CAP ? count : AG_SCR_CAP);
  }
  // grid lines: the pixel column / row a line falls into (one pixel wide), and which columns / rows lie inside the arena
  const float sx_scale = (float)o.W * 0.5f / half_w, sy_scale = (float)o.H * 0.5f / half_h, spacing = Wd / 7.0f;
  // (r05: the eight lines' pixel columns / rows are eight numbers -- every thread takes them from the same expression and compares; each column / row
  // used to evaluate the eight floors for itself)
  int gcol[8], grow[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { const float g = (float)i * spacing; gcol[i] = (int)floorf((g - px) * sx_scale + (float)o.W * 0.5f); grow[i] = (int)floorf((g - py) * sy_scale + (float)o.H * 0.5f); }
  for (int k = (int)threadIdx.x; k < o.W; k += 256) {
    const float wx = px + (((float)k + 0.5f) / (float)o.W * 2.0f - 1.0f) * half_w;
    uint8_t f = (wx >= 0.0f && wx <= Wd) ? 2 : 0;
#pragma unroll
    for (int i = 0; i < 8; i++) if (gcol[i] == k) f |= 1;
    colflag[k] = f; colx[k] = wx;
  }
  for (int k = (int)threadIdx.x; k < o.H; k += 256) {
    const float wy = py + (((float)k + 0.5f) / (float)o.H * 2.0f - 1.0f) * half_h;
    uint8_t f = (wy >= 0.0f && wy <= Wd) ? 2 : 0;
#pragma unroll
    for (int i = 0; i < 8; i++) if (grow[i] == k) f |= 1;
    rowflag[k] = f; rowy[k] = wy;
  }
  __syncthreads();
  for (int k = (int)threadIdx.x; k < n_list; k += 256) {   // conservative pixel box of every listed entity (one pixel of margin: the inside test decides)
    const float x = ex[k], y = ey[k], r = er[k];
    int c0 = (int)floorf(((x - r - px) / half_w + 1.0f) * 0.5f * (float)o.W - 0.5f) - 1, c1 = (int)floorf(((x + r - px) / half_w + 1.0f) * 0.5f * (float)o.W - 0.5f) + 2;
    int r0 = (int)floorf(((y - r - py) / half_h + 1.0f) * 0.5f * (float)o.H - 0.5f) - 1, r1 = (int)floorf(((y + r - py) / half_h + 1.0f) * 0.5f * (float)o.H - 0.5f) + 2;
    c0 = c0 < 0 ? 0 : c0; c1 = c1 > o.W - 1 ? o.W - 1 : c1; r0 = r0 < 0 ? 0 : r0; r1 = r1 > o.H - 1 ? o.H - 1 : r1;
    if (c0 > c1 || r0 > r1) { c0 = 1; c1 = 0; r0 = 1; r1 = 0; }
    ebx[k] = (unsigned)c0 | ((unsigned)c1 << 16); eby[k] = (unsigned)r0 | ((unsigned)r1 << 16);
    eapo[k] = r * scr_cos_half_step((int)(ec[k] >> 24));
  }
  __syncthreads();
  // (wave-uniform values the compiler cannot know to be uniform -- the wavefront's number, what comes out of LDS at a uniform address -- go through
  // v_readfirstlane: the painter's loops over entities, boxes and tiles are then scalar loops, not vector compares and exec masks)
  const int n = __builtin_amdgcn_readfirstlane(n_list), wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int band_rows = AG_SCR_BAND / o.W > 0 ? (AG_SCR_BAND / o.W < o.H ? AG_SCR_BAND / o.W : o.H) : 1;
  int a1 = 255, a2 = 255;   // agent view: final alphas of the two previous pixels (thread 0 carries them from band to band)
  for (int row0 = 0; row0 < o.H; row0 += band_rows) {   // row 0 = bottom (glReadPixels)
    const int rows = o.H - row0 < band_rows ? o.H - row0 : band_rows, npix = rows * o.W;
    // background + grid
    constexpr unsigned GRIDV = AGV ? 0x1A000000u : 0xFF00001Au, BACKV = AGV ? 0u : 0x00FFFFFFu;
    // (0.1, 0, 0) -> 26; alpha byte: a fragment was written.  Agent view (r05): the pixel is written in the form post_processing_frame_data leaves
    // it in -- a value <= 230 moves into alpha and the channel is cleared, whatever lies around it -- so that no pass over the band has to do it
    if (SCR_ABL(16)) {} else
    if ((o.W & 3) == 0) {   // four pixels of a row per lane and store (r05): one row flag, the four column flags as one word, a 16-byte LDS store
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const int gpr = o.W >> 2, ngrp = npix >> 2;   // groups per row, groups in the band
      const int step_r = 256 / gpr, step_c = 256 - step_r * gpr;
      for (int g = (int)threadIdx.x, r = (int)threadIdx.x / gpr, gc = (int)threadIdx.x - r * gpr; g < ngrp; g += 256, r += step_r, gc += step_c) {
        if (gc >= gpr) { gc -= gpr; r += 1; }
        const unsigned cf4 = *(const unsigned *)&colflag[gc << 2]; const unsigned rf = rowflag[row0 + r];
        const unsigned line = (rf & 2u) ? 0x01010101u : 0u, inside = (rf & 1u) ? 0x02020202u : 0u;     // a column's line shows in rows inside the arena; a row's line in columns inside
        const unsigned m = (cf4 & line) | ((cf4 & inside) >> 1);                                        // byte i: pixel i of the group is a grid pixel
        v4u v; v.x = (m & 0x01u) ? GRIDV : BACKV; v.y = (m & 0x0100u) ? GRIDV : BACKV; v.z = (m & 0x010000u) ? GRIDV : BACKV; v.w = (m & 0x01000000u) ? GRIDV : BACKV;
        *(v4u *)&fb[g << 2] = v;
      }
    } else {
    const int step_r = 256 / o.W, step_c = 256 - step_r * o.W;   // (one division: the loop below walks rows and columns by addition)
    for (int q = (int)threadIdx.x, r = (int)threadIdx.x / o.W, cidx = (int)threadIdx.x - r * o.W; q < npix; q += 256, r += step_r, cidx += step_c) {
      if (cidx >= o.W) { cidx -= o.W; r += 1; }
      const uint8_t cf = colflag[cidx], rf = rowflag[row0 + r];
      const bool grid = ((cf & 1) && (rf & 2)) || ((rf & 1) && (cf & 2));
      fb[q] = grid ? GRIDV : BACKV;
    }
    }
    if (threadIdx.x == 0) pp_chunks = 0ull;
    __syncthreads();
    // entities in draw order; a wavefront paints only its own rows, so later entities overwrite earlier ones without any exchange
    const int rpw = (rows + 3) >> 2, wr0 = row0 + wave * rpw, wr1 = (wr0 + rpw < row0 + rows ? wr0 + rpw : row0 + rows) - 1;
    unsigned long long mybits = 0ull;   // agent view: chunks this wavefront painted 255-pixels into
    // (r05: which entities reach into this wavefront's rows is decided for 64 entities at a time, a lane each, and only those are visited -- in
    // list order, so later draws still overwrite earlier ones.  Walking the whole list with a uniform box test per entity, wavefront and band was
    // a third of the kernel's instructions at 128 x 128, where a frame is five bands and a wavefront owns 7 of its rows: ~3 of ~40 entities hit.)
    if (wr0 <= wr1 && !SCR_ABL(2)) for (int k0 = 0; k0 < n; k0 += 64) {
      unsigned long long hits;
      { const int kk = k0 + lane; const bool in_ = kk < n;
        const unsigned bx_ = in_ ? ebx[kk] : 1u, by_ = in_ ? eby[kk] : 1u;   // (1 = first 1, last 0: empty)
        const int r0_ = (int)(by_ & 0xFFFFu), r1_ = (int)(by_ >> 16);
        hits = __ballot(in_ && (int)(bx_ & 0xFFFFu) <= (int)(bx_ >> 16) && (r0_ < wr0 ? wr0 : r0_) <= (r1_ > wr1 ? wr1 : r1_)); }
      for (; hits; hits &= hits - 1ull) {
      const int k = k0 + (int)__builtin_ctzll(hits);
      const unsigned bx = (unsigned)__builtin_amdgcn_readfirstlane((int)ebx[k]), by = (unsigned)__builtin_amdgcn_readfirstlane((int)eby[k]);
      const int c0 = (int)(bx & 0xFFFFu), c1 = (int)(bx >> 16);
      int r0 = (int)(by & 0xFFFFu), r1 = (int)(by >> 16);
      r0 = r0 < wr0 ? wr0 : r0; r1 = r1 > wr1 ? wr1 : r1;
      const float x = ex[k], y = ey[k], r = er[k], apo = eapo[k]; const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)ec[k]);
      // agent view: the main agent's 230 is written as post-processed (alpha 230, no colour); a 255-colour stays a 255-pixel, and the 64-pixel
      // chunks of the band it may fall into are marked for the run pass below (a superset: a later draw may paint over it)
      const bool keep255 = CH == 4 && (e & 0xFFFFFFu) > 230u;
      const unsigned paint = (CH == 4 && !keep255) ? ((e & 0xFFu) << 24) : ((e & 0xFFFFFFu) | 0xFF000000u);
      if (keep255) {   // (wave-uniform arithmetic: every chunk from the box's first pixel to its last -- a superset of the chunks it touches)
        const int lo = ((r0 - row0) * o.W + c0) >> 6, hi = ((r1 - row0) * o.W + c1) >> 6;
        mybits |= ((hi - lo >= 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull)) << lo;
      }
      for (int ty = r0; ty <= r1; ty += 8) for (int tx = c0; tx <= c1; tx += 8) {   // 8 x 8 pixel tiles of the box, a lane per pixel
        const int rr = ty + (lane >> 3), cc = tx + (lane & 7);
        if (rr <= r1 && cc <= c1 && scr_inside_apo(colx[cc] - x, rowy[rr] - y, r, apo, (int)(e >> 24))) fb[(rr - row0) * o.W + cc] = paint;
      }
      }
    }
    if (CH == 4 && lane == 0 && mybits) atomicOr(&pp_chunks, mybits);
    __syncthreads();
    if (CH == 4 && !SCR_ABL(4)) {  // ScreenObservation::post_processing_frame_data (ScreenEnvironment.hpp:48-88): a sequential pass over the flat buffer
      // With the colours this kernel paints (one non-zero channel per pixel: 26 grid, 230 main agent, 255 pellets / others / viruses) the
      // pass has a closed form.  A pixel whose channel is <= 230 moves it into alpha: no dependence -- since r05 the painter writes such pixels
      // in that form at once (the kernel is bound by issued instructions: a pass over every pixel of the band cost as much as painting it).
      // A 255-pixel keeps alpha 255 unless the two previous FINAL alphas are both <= 30, then it takes the previous one; so a run of consecutive
      // 255-pixels takes ONE value, decided at its first pixel -- 255, or the alpha in front of the run -- and only run starts are sequential: a
      // dozen per frame instead of 7056 pixels.  One wavefront walks them in pixel order (on four wavefronts the pass issued more instructions
      // and was slower: 352 -> 380 us per 4096 frames of 128 x 128), and since r05 only through the 64-pixel chunks the painter marked as
      // holding a 255-pixel (pp_chunks) instead of through every chunk of the band.
      // All four wavefronts share the pass (r05): the kernel is bound by what a SIMD issues and a workgroup's wavefronts sit on different SIMDs, so a
      // pass on wavefront 0 alone is issued by one SIMD in four.  The band's chunks are cut into four runs of consecutive chunks at SAFE boundaries --
      // the two pixels in front of the boundary carry no colour: no run crosses it and a start behind it reads final alphas -- found with one ballot
      // (lane c looks at the two pixels in front of chunk c); every wavefront walks its chunks in order exactly as the single wavefront did.
      {
        const int nch = (npix + 63) >> 6, per = (nch + 3) >> 2;
        bool sf = false;
        if (lane >= 1 && lane < nch) sf = (fb[64 * lane - 1] & 0xFFFFFFu) == 0u && (fb[64 * lane - 2] & 0xFFFFFFu) == 0u;
        const unsigned long long safe = __ballot(sf);
        auto cut = [&](int t) -> int { if (t <= 0) return 0; if (t >= nch) return nch; const unsigned long long m = safe >> t; return m ? t + (int)__builtin_ctzll(m) : nch; };   // first safe boundary >= t
        const int lo_c = cut(wave * per), hi_c = cut((wave + 1) * per);
        const unsigned long long seg = (hi_c >= 64 ? ~0ull : ((1ull << hi_c) - 1ull)) & ~((1ull << lo_c) - 1ull);
        int run_val = -1, prev_c = -2;   // value of the run that reaches into the current chunk from the left (-1: none); the chunk visited before this one
        const unsigned long long lt_lane = (1ull << lane) - 1ull;
        // (pp_chunks comes out of LDS: a vector register to the compiler, which then walked the chunks with vector compares and exec masks)
        const unsigned long long cm0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pp_chunks >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pp_chunks);
        for (unsigned long long cm = cm0 & seg; cm; cm &= cm - 1ull) {
          const int cch = (int)__builtin_ctzll(cm), c0_ = cch << 6;
          if (cch != prev_c + 1) run_val = -1;                     // (the chunks in between hold no 255-pixel: no run crosses them)
          prev_c = cch;
          const int q = c0_ + lane; const bool in = q < npix; const int qi = in ? q : 0;
          // the chunk's words and, for the lanes that turn out to start a run, the two pixels in front: three reads in one round trip, no branches
          const unsigned w = fb[qi], wl1 = fb[qi >= 1 ? qi - 1 : 0], wl2 = fb[qi >= 2 ? qi - 2 : 0];
          const bool X = in && (w & 0xFFFFFFu) != 0u;            // only 255-pixels carry a colour
          unsigned long long xm = __ballot(X);
          if (xm == 0ull) { run_val = -1; continue; }
          // All runs of the chunk at once (r05): a 255-pixel finds the start of its run in the ballot (the highest non-255 pixel below it), the
          // start decides the run's value from the two pixels in front of it -- final already unless the second one is a 255-pixel of THIS chunk
          // (two runs one pixel apart: the chunk then takes the run-by-run loop below) -- and hands it to the run with one ds_bpermute.
          // (Measured: 128 x 128 x 4 on a task-3 state 203 -> 199 us; marking the chunks row by row so that fewer are visited: 413 us on task 1's
          // pattern -- the painter's scalar instructions --; neighbours from registers instead of LDS: 218 us.  The kernel is bound by what it issues.)
          if (!((xm & ~(xm << 1)) & (xm << 2))) {   // (no two runs one pixel apart -- X[i], !X[i-1], X[i-2] --: scalar arithmetic on the ballot)
            const unsigned long long below = ~xm & lt_lane;
            const int sp = below ? 64 - (int)__builtin_clzll(below) : 0;            // chunk position at which this lane's run starts (0: it reaches the left edge)
            const bool from_left = !below && run_val >= 0;                           // ... and continues the run of the chunk before
            const bool start = X && sp == lane && !from_left;
            const int f1 = q >= 1 ? (int)(wl1 >> 24) : a1, f2 = q >= 2 ? (int)(wl2 >> 24) : (q == 1 ? a1 : a2);
            const int val = (start && row0 * o.W + q >= 2 && f2 <= 30 && f1 <= 30) ? f1 : 255;
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
              const int p = row0 * o.W + c0_ + s_;              // flat pixel index of the run start
              int f1, f2;                                        // final alphas of pixels p - 1 and p - 2
              if (c0_ + s_ >= 1) f1 = (int)(fb[c0_ + s_ - 1] >> 24); else f1 = a1;
              if (c0_ + s_ >= 2) f2 = (int)(fb[c0_ + s_ - 2] >> 24); else f2 = (c0_ + s_ == 1) ? a1 : a2;
              val = (p >= 2 && f2 <= 30 && f1 <= 30) ? f1 : 255;
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
      }
      __syncthreads();
      // the band's last two final alphas, for the next band
      { const int l1 = (int)(fb[npix - 1] >> 24), l2 = npix >= 2 ? (int)(fb[npix - 2] >> 24) : a1; a2 = l2; a1 = l1; }
    }
    // the band leaves LDS as one coalesced byte stream
    uint8_t *bd = dst + (size_t)row0 * o.W * CH; const int nbytes = npix * CH;
    // (r05: whole 32-bit words instead of one byte per lane and store -- a 128 x 128 x 4 frame left as 256 byte-stores per thread.  A packed RGBA
    // pixel IS its four output bytes (0xAABBGGRR, little endian); three-channel frames assemble each output word from the two pixels it spans.
    // Frames and bands start on 4-byte boundaries when their byte counts say so; anything else takes the byte loop.)
    // (r05, second pass: four pixels per lane -- one 16-byte LDS read; the agent view stores them as they are, 16 bytes; a three-channel frame packs
    // them into three words instead of assembling every output word with a division by three)
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    if (SCR_ABL(8)) {} else
    if (CH == 4 && (((size_t)bd) & 15) == 0 && (npix & 3) == 0) { v4u_ *bw = (v4u_ *)bd; const v4u_ *fw = (const v4u_ *)fb; for (int g = (int)threadIdx.x; g < (npix >> 2); g += 256) bw[g] = fw[g]; }
    else if (CH == 3 && (((size_t)bd) & 3) == 0 && (npix & 3) == 0) {
      unsigned *bw = (unsigned *)bd; const v4u_ *fw = (const v4u_ *)fb;
      for (int g = (int)threadIdx.x; g < (npix >> 2); g += 256) {
        const v4u_ v = fw[g]; const unsigned p0 = v.x & 0xFFFFFFu, p1 = v.y & 0xFFFFFFu, p2 = v.z & 0xFFFFFFu, p3 = v.w & 0xFFFFFFu;
        bw[3 * g] = p0 | (p1 << 24); bw[3 * g + 1] = (p1 >> 8) | (p2 << 16); bw[3 * g + 2] = (p2 >> 16) | (p3 << 8);
      }
    }
    else if (CH == 4 && (((size_t)bd) & 3) == 0) { unsigned *bw = (unsigned *)bd; for (int q = (int)threadIdx.x; q < npix; q += 256) bw[q] = fb[q]; }
    else if (CH == 3 && (((size_t)bd) & 3) == 0) {
      unsigned *bw = (unsigned *)bd; const int nwords = nbytes >> 2;
      for (int j = (int)threadIdx.x; j < nwords; j += 256) {
        // output bytes 4j .. 4j+3 = channels (4j + t) % 3 of pixels (4j + t) / 3: 4j = 3 q0 + c0
        const int b0 = 4 * j, q0 = b0 / 3, c0 = b0 - 3 * q0;
        // (c0 <= 2: the word's last byte is byte 5 of the two pixels' six)
        const unsigned long long two = (unsigned long long)(fb[q0] & 0xFFFFFFu) | ((unsigned long long)(fb[q0 + 1 < npix ? q0 + 1 : q0] & 0xFFFFFFu) << 24);
        bw[j] = (unsigned)(two >> (8 * c0));
      }
      for (int b = 4 * nwords + (int)threadIdx.x; b < nbytes; b += 256) { const int q = b / 3, ch = b - q * 3; bd[b] = (uint8_t)((fb[q] >> (8 * ch)) & 0xFFu); }
    }
    else if (CH == 4) { for (int b = (int)threadIdx.x; b < nbytes; b += 256) bd[b] = (uint8_t)((fb[b >> 2] >> (8 * (b & 3))) & 0xFFu); }
    else { for (int b = (int)threadIdx.x; b < nbytes; b += 256) { const int q = b / 3, ch = b - q * 3; bd[b] = (uint8_t)((fb[q] >> (8 * ch)) & 0xFFu); } }
    __syncthreads();
  }
}
#endif
