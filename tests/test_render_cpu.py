"""Headless renderer (render.py; SURVEY 8(f) f4, optional item): the scene of fortattack.py:368-600 rasterised with
numpy.  CPU: pixel-level expectations on a hand-built state."""
import numpy as np

from emergent_multiagent_strategies_amd.render import render_frame, render_state


def _px(img, x, y):
    """pixel holding world point (x, y), camera bounds +-1, row 0 at the top"""
    size = img.shape[0]
    return img[int((1.0 - (y + 1.0) / 2.0) * size), int((x + 1.0) / 2.0 * size)]


def test_scene_layout_and_agents():
    px = np.array([-0.5, 0.0, 0.5, 0.0])
    py = np.array([-0.5, -0.2, -0.5, 0.4])
    ang = np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2])
    alive = np.array([1, 1, 0, 1])
    img = render_frame(px, py, ang, alive, num_guards=2, size=400)
    assert img.shape == (400, 400, 3) and img.dtype == np.uint8
    assert tuple(_px(img, 0.9, 0.0)) == (0, 0, 0)                   # active region is black
    assert tuple(_px(img, 0.0, 0.9)) == (128, 128, 128)             # grey strip above the wall ...
    assert tuple(_px(img, 0.0, -0.9)) == (128, 128, 128)            # ... and below
    assert tuple(_px(img, 0.1, 0.75)) == (0, 255, 255)              # the fort circle at the door (0, 0.8)
    assert tuple(_px(img, -0.5, -0.5)) == (0, 255, 0)               # guard 0
    assert tuple(_px(img, 0.0, -0.2)) == (0, 255, 0)                # guard 1
    assert tuple(_px(img, 0.0, 0.4)) == (255, 0, 0)                 # attacker 3
    assert tuple(_px(img, 0.5, -0.5)) == (0, 0, 0)                  # the dead attacker is not drawn ...
    dead = render_frame(px, py, ang, alive, num_guards=2, size=400, viz_dead=True)
    assert dead[..., 0].astype(int)[int((1 - 0.25) * 400), int(0.75 * 400)] > 60   # ... unless asked for (translucent)
    # the head disc sits towards the heading: guard 0 looks along +x
    assert tuple(_px(img, -0.5 + 0.05 + 0.01, -0.5)) == (0, 255, 0) and tuple(_px(img, -0.5 - 0.05 - 0.01, -0.5)) == (0, 0, 0)


def test_laser_triangle_matches_the_reference_wedge():
    """core.py:373-382: apex at pos + size (cos a, sin a), two far corners at shootRad under +- shootWin / 2."""
    img = render_frame([0.0, 0.0], [-0.5, 0.7], [np.pi / 2, 0.0], [1, 1], num_guards=1, shoot=[True, False], size=500)
    inside, outside = _px(img, 0.0, -0.1), _px(img, 0.45, -0.3)
    assert inside[1] > 60 and inside[0] == 0 and tuple(outside) == (0, 0, 0)   # translucent green inside the wedge only
    far = _px(img, 0.0, -0.5 + 0.05 + 0.8 * np.cos(np.pi / 8) + 0.03)          # beyond the far edge
    assert tuple(far) == (0, 0, 0)
    none = render_frame([0.0, 0.0], [-0.5, 0.7], [np.pi / 2, 0.0], [1, 1], num_guards=1, size=500)
    assert tuple(_px(none, 0.0, -0.1)) == (0, 0, 0)


def test_render_state_picks_one_env_of_a_batch():
    st = dict(pos_x=np.array([[0.0, 0.3], [-0.6, 0.6]]), pos_y=np.array([[0.0, 0.3], [-0.6, -0.6]]),
              ang=np.zeros((2, 2)), alive=np.ones((2, 2), np.uint8))
    a, b = render_state(st, 0, num_guards=1, size=200), render_state(st, 1, num_guards=1, size=200)
    assert tuple(_px(a, 0.0, 0.0)) == (0, 255, 0) and tuple(_px(b, 0.0, 0.0)) == (0, 0, 0)
    assert tuple(_px(b, 0.6, -0.6)) == (255, 0, 0)
