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
// draw order (ordered ballot compaction); then every thread shades pixels, walking the list and keeping the last hit.
#pragma once
#include "agar_types.h"

#define AG_SCR_CAP 512  // visible entities kept per frame (more are dropped from the END of the draw order)

struct AgScreenCfg { int W, H, agent_view; };  // agent_view: the 4-channel frame of Renderer::multi_channel_render_screen

#ifndef AGAR_CPU_EMU
__device__ __forceinline__ unsigned scr_palette(int k) {  // core/color.hpp:4-12 as 0xBBGGRR bytes (GL rounds c * 255 to nearest)
  const unsigned pal[6] = {0x0000FFu /*red*/, 0x00A6FFu /*orange 1,.65,0*/, 0x00FFFFu /*yellow*/, 0x00FF00u /*green*/, 0xFF0000u /*blue*/, 0xCC3399u /*purple .6,.2,.8*/};
  return pal[((k % 6) + 6) % 6];
}
__device__ __forceinline__ bool scr_inside(float dx, float dy, float r, int nsides) {
  float d2 = dx * dx + dy * dy;
  if (d2 > r * r) return false;
  const float step = 6.28318530717958647692f / (float)nsides;
  float apo = r * cosf(0.5f * step);
  if (d2 <= apo * apo) return true;
  float th = atan2f(dy, dx); if (th < 0.0f) th += 6.28318530717958647692f;
  float k = floorf(th / step);
  float phi = (k + 0.5f) * step;
  return dx * cosf(phi) + dy * sinf(phi) <= apo;
}

__global__ void __launch_bounds__(256) k_screen_obs(const AgState *__restrict__ gs, AgScreenCfg o, uint8_t *out) {
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
#endif
