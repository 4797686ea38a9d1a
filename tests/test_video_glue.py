"""render() / video recorder glue of the gym wrapper (N4, /root/reference/gym_agario/AgarioEnv.py:134-181,366-404)."""
import os
import struct

import numpy as np
import pytest


def test_mjpeg_avi_container(tmp_path):
    from agarcl_amd.video import write_mjpeg_avi
    frames = [np.full((84, 84, 3), 20 * i, np.uint8) for i in range(10)]
    path = str(tmp_path / "v.avi")
    how = write_mjpeg_avi(path, frames, fps=60.0)
    b = open(path, "rb").read()
    if how == "builtin":
        assert b[:4] == b"RIFF" and b[8:12] == b"AVI " and struct.unpack("<I", b[4:8])[0] == len(b) - 8
        assert b.count(b"00dc") == 2 * len(frames)                       # one data chunk + one index entry per frame
        i = b.index(b"avih"); us, _, _, flags, total = struct.unpack("<IIIII", b[i + 8:i + 28])
        assert us == 16667 and flags & 0x10 and total == len(frames)
        j = b.index(b"00dc"); n = struct.unpack("<I", b[j + 4:j + 8])[0]
        assert b[j + 8:j + 10] == b"\xff\xd8"                              # a JPEG stream
        from PIL import Image
        import io
        im = Image.open(io.BytesIO(b[j + 8:j + 8 + n])); assert im.size == (84, 84)
    assert os.path.getsize(path) > 1000


def test_color_names_match_the_reference_values():
    from agarcl_amd.agar_utils import Color, get_color_array
    assert [c.name for c in Color] == ["RED", "ORANGE", "YELLOW", "GREEN", "BLUE", "PURPLE", "WHITE", "BLACK", "LAST"]
    assert get_color_array(Color.PURPLE).tolist() == [0.6 * 255, 0.2 * 255, 0.8 * 255] and get_color_array(Color.GREEN).tolist() == [0, 255, 0]
    with pytest.raises(ValueError):
        get_color_array(Color.LAST)


@pytest.mark.gpu
@pytest.mark.parametrize("obs_type,kw", [("screen", dict(agent_view=True)), ("screen", dict()), ("grid", dict())], ids=["agent_view", "screen", "grid"])
def test_render_and_video_recorder_gpu(tmp_path, obs_type, kw):
    from agarcl_amd.gym_agario import AgarioEnv
    env = AgarioEnv(obs_type=obs_type, render_mode="rgb_array", num_viruses=5, **kw)
    env.seed(4); env.reset()
    env.enable_video_recorder()
    for t in range(6):
        obs, r, done, trunc, info = env.step(((0.3, -0.2), 0))
    fr = env.render()
    if obs_type == "screen":
        assert fr is obs
    else:
        assert fr.shape == (1, 512, 512, 3) and fr.dtype == np.uint8 and fr.min() < 255     # get_frame(): something is drawn
    assert len(env.video_recorder) == 6
    v = env.video_recorder[0]
    if obs_type == "grid":
        assert v.shape == (512, 512, 3)
    elif kw.get("agent_view"):
        assert v.shape == (84, 84, 3) and set(map(tuple, v.reshape(-1, 3))) <= {(255, 0, 0), (255, 255, 255), (153, 51, 204), (0, 255, 0), (0, 0, 255), (26, 0, 0)}
    path = env.generate_video(str(tmp_path), "run.avi")
    assert path and os.path.getsize(path) > 2000
    env.disable_video_recorder()
    assert env.generate_video(str(tmp_path), "no.avi") is None
    env.close()
