"""Colour names of the gym wrapper's video frames (the reference's gym_agario/agar_utils.py: same enum members and RGB values, because
recorded agent-view videos are painted with them)."""
import enum
import random

import numpy as np


class Color(enum.Enum):
    RED = 1; ORANGE = 2; YELLOW = 3; GREEN = 4; BLUE = 5; PURPLE = 6; WHITE = 7; BLACK = 8; LAST = 9


_RGB = {Color.RED: (1.0, 0.0, 0.0), Color.ORANGE: (1.0, 0.65, 0.0), Color.YELLOW: (1.0, 1.0, 0.0), Color.GREEN: (0.0, 1.0, 0.0),
        Color.BLUE: (0.0, 0.0, 1.0), Color.PURPLE: (0.6, 0.2, 0.8), Color.WHITE: (1.0, 1.0, 1.0), Color.BLACK: (0.0, 0.0, 0.0)}


def get_color_array(c):
    if c not in _RGB:
        raise ValueError("Not a color")
    return np.array(_RGB[c]) * 255


def random_color():
    return get_color_array(Color(random.randint(1, Color.LAST.value - 1)))
