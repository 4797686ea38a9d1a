"""TEST INFRASTRUCTURE: numpy restatement of the reference's screen frame as rasterisation rules.
PARITY UNPINNED -- the reference draws with OpenGL (agario/rendering/renderer.hpp:91-185, core/renderables.hpp,
rendering/FrameBufferObject.hpp:105) and no GL context exists in the build container, so nothing here could be checked
against the reference's pixels; the rules are restated from its source:
  camera z = clamp(100 + mass/10, 100, 900) above the player's centre, 45 deg vertical FOV (renderer.hpp:91-120);
  white background, then 8x8 grid lines (0.1,0,0), pellets, foods, players in map order, viruses (renderer.hpp:163-185);
  regular polygons with vertices at k*2pi/N, N = 5 / 7 / 50 / 150 (renderables.hpp:198-207, Entities.hpp:13-16);
  glReadPixels rows bottom-up, RGB bytes.
Colours of pellets / foods / agents are rand()-chosen in the reference's renderable build and not reproducible; like the
product this uses palette[id % 6] / palette[pid % 6]."""
import numpy as np

from . import blob

PALETTE = np.array([[255, 0, 0], [255, 166, 0], [255, 255, 0], [0, 255, 0], [0, 0, 255], [153, 51, 204]], dtype=np.uint8)
BOT_COLOR = {1: 4, 2: 5, 3: 0, 4: 1}


def radius(mass):
    return np.float32(np.sqrt(np.float64(mass) / 1.0 / np.pi))


def _inside(dx, dy, r, n):
    f = np.float32
    d2 = dx * dx + dy * dy
    step = f(6.28318530717958647692) / f(n)
    apo = r * np.cos(f(0.5) * step, dtype=np.float32)
    th = np.arctan2(dy, dx, dtype=np.float32)
    th = np.where(th < 0, th + f(6.28318530717958647692), th).astype(np.float32)
    k = np.floor(th / step).astype(np.float32)
    phi = ((k + f(0.5)) * step).astype(np.float32)
    proj = dx * np.cos(phi, dtype=np.float32) + dy * np.sin(phi, dtype=np.float32)
    return (d2 <= r * r) & ((d2 <= apo * apo) | (proj <= apo))


def render(blob_words, arena_size, agent_pid, kinds, W=84, H=84, agent_view=False, main_pid=None):
    """uint8 [H][W][3] (or [H][W][4] with agent_view: Renderer::multi_channel_render_screen, renderer.hpp:128-155, followed
    by ScreenObservation::post_processing_frame_data, ScreenEnvironment.hpp:48-88), rows bottom-up.
    kinds: AG_KIND_* per player in the blob's (map iteration) order; main_pid: state.main_agent_pid (the last agent added)."""
    f = np.float32
    d = blob.parse(blob_words)
    pl = [p for p in d["players"] if p["pid"] == agent_pid][0]
    cx = pl["cell_f"][:, 0].astype(np.float32); cy = pl["cell_f"][:, 1].astype(np.float32); cm = pl["cell_mass"].astype(np.int64)
    sx = f(0); sy = f(0); tm = 0
    for x, y, m in zip(cx, cy, cm):       # Player::x/y: sequential fp32 sums in cell order (core/Player.hpp:102-126)
        sx = f(sx + f(x * f(m))); sy = f(sy + f(y * f(m))); tm += int(m)
    px, py = f(sx / f(tm)), f(sy / f(tm))
    z = f(min(max(100.0 + tm / 10.0, 100.0), 900.0))
    half_h = f(z * f(0.41421356237309504880)); half_w = f(half_h * (f(W) / f(H)))
    cols = np.arange(W, dtype=np.float32); rows = np.arange(H, dtype=np.float32)
    wx = (px + ((cols + f(0.5)) / f(W) * f(2) - f(1)) * half_w).astype(np.float32)[None, :].repeat(H, 0)
    wy = (py + ((rows + f(0.5)) / f(H) * f(2) - f(1)) * half_h).astype(np.float32)[:, None].repeat(W, 1)
    img = np.full((H, W, 3), 255, dtype=np.uint8) if not agent_view else np.zeros((H, W, 3), dtype=np.uint8)
    drawn = np.zeros((H, W), dtype=bool)
    Wd = f(arena_size); spacing = f(Wd / f(7))
    in_x = (wx >= 0) & (wx <= Wd); in_y = (wy >= 0) & (wy <= Wd)
    sxs = f(f(W) * f(0.5) / half_w); sys_ = f(f(H) * f(0.5) / half_h)
    ci = np.arange(W)[None, :].repeat(H, 0); ri = np.arange(H)[:, None].repeat(W, 1)
    for i in range(8):
        g = f(f(i) * spacing)
        gc = int(np.floor(f(f(g - px) * sxs + f(W) * f(0.5)))); gr = int(np.floor(f(f(g - py) * sys_ + f(H) * f(0.5))))
        gm = ((ci == gc) & in_y) | ((ri == gr) & in_x)
        img[gm] = (26, 0, 0); drawn[gm] = True

    def draw(x, y, r, n, color):
        x = f(x); y = f(y); r = f(r)
        if abs(x - px) > half_w + r or abs(y - py) > half_h + r:
            return
        m = _inside((wx - x).astype(np.float32), (wy - y).astype(np.float32), r, n)
        img[m] = color; drawn[m] = True
    r_pel, r_food = radius(1), radius(10)
    for x, y, i in zip(d["pellet_x"], d["pellet_y"], d["pellet_id"]):
        draw(x, y, r_pel, 5, (255, 0, 0) if agent_view else PALETTE[int(i) % 6])
    for x, y, i in zip(d["food_x"], d["food_y"], d["food_id"]):
        draw(x, y, r_food, 7, (255, 0, 0) if agent_view else PALETTE[int(i) % 6])
    order = list(zip(d["players"], kinds))
    if agent_view:   # the main agent first (type 3), then everybody else (type 1)
        order = [pk for pk in order if pk[0]["pid"] == main_pid] + [pk for pk in order if pk[0]["pid"] != main_pid]
    for p, kind in order:
        if agent_view:
            color = (230, 0, 0) if p["pid"] == main_pid else (0, 255, 0)
        else:
            color = PALETTE[p["pid"] % 6] if kind == 0 else PALETTE[BOT_COLOR[kind]]
        for (x, y), m in zip(p["cell_f"][:, :2], p["cell_mass"]):
            draw(x, y, radius(int(m)), 50, color)
    for x, y, m in zip(d["virus_x"], d["virus_y"], d["virus_mass"]):
        draw(x, y, radius(int(m)), 150, (0, 0, 255) if agent_view else PALETTE[3])
    if not agent_view:
        return img
    data = np.concatenate([img, np.where(drawn, 255, 0).astype(np.uint8)[:, :, None]], axis=2).reshape(-1).copy()
    n = W * H * 4
    for i in range(n):   # ScreenObservation::post_processing_frame_data, byte for byte (ScreenEnvironment.hpp:48-88)
        if i % 4 != 3 and data[i] != 0:
            if data[i] <= 230:
                data[i + (3 - i % 4)] = data[i]; data[i] = 0
            else:
                prev = i - i % 4 - 1; prev2 = prev - 4
                if prev2 >= 0 and data[prev2] <= 30 and data[prev] <= 30:
                    data[i + (3 - i % 4)] = data[prev]
    return data.reshape(H, W, 4)
