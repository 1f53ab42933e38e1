"""Headless frame renderer: one env's SoA state -> an RGB image (optional item).

The reference draws its scene with pyglet in a 700 x 700 window (gym_fortattack/fortattack.py:368-600: black
active region between the walls, the cyan fort circle at the door, every agent a disc in its team colour with a
smaller "head" disc towards its heading, a translucent laser triangle for an agent that shoots, grey strips above
and below the active region; camera bounds +-1).  This is the same picture rasterised with numpy -- no window, no
GL -- for evaluation videos of thousands of GPU-side envs: `render_frame` takes plain arrays (a row of
`BatchedFortAttack.get_state()`), `FortAttackGlobalEnv.render(mode="rgb_array")` calls it for the facade's env.
It is not on the hot path and nothing in the engine depends on it.
"""
import numpy as np

GUARD_RGB = (0.0, 1.0, 0.0)       # fortattack_env_v1.py:57
ATTACKER_RGB = (1.0, 0.0, 0.0)
FORT_RGB = (0.0, 1.0, 1.0)        # fortattack.py:432
STRIP_RGB = (0.5, 0.5, 0.5)


def _blend(img, mask, rgb, alpha=1.0):
    c = np.asarray(rgb, np.float32)
    img[mask] = img[mask] * (1.0 - alpha) + c * alpha


def render_frame(pos_x, pos_y, ang, alive, num_guards, shoot=None, size=350, viz_dead=False,
                 wall_pos=(-1.0, 1.0, -0.8, 0.8), agent_size=0.05, fort_dim=0.15, door=(0.0, 0.8),
                 shoot_rad=0.8, shoot_win=np.pi / 4):
    """pos_x, pos_y, ang, alive: (N,) arrays of ONE env (guards first); shoot: (N,) bool, the agents whose last
    action was `shoot` (laser triangles, core.py:373-382).  Returns uint8 (size, size, 3), y up."""
    px, py, ang = (np.asarray(v, np.float64).reshape(-1) for v in (pos_x, pos_y, ang))
    alive = np.asarray(alive).reshape(-1) != 0
    n = px.shape[0]
    shoot = np.zeros(n, bool) if shoot is None else np.asarray(shoot).reshape(-1).astype(bool)
    # pixel centres in world coordinates, camera bounds +-1 (fortattack.py:582-587); row 0 = top
    c = (np.arange(size) + 0.5) / size * 2.0 - 1.0
    X, Y = np.meshgrid(c, -c)
    img = np.full((size, size, 3), 1.0, np.float32)
    xmin, xmax, ymin, ymax = wall_pos
    inside = (X >= xmin) & (X <= xmax)
    _blend(img, inside & (Y >= ymin) & (Y <= ymax), (0.0, 0.0, 0.0))          # active region
    _blend(img, (X - door[0]) ** 2 + (Y - door[1]) ** 2 <= fort_dim ** 2, FORT_RGB)
    for i in range(n):                                                         # lasers under the agents
        if not (alive[i] and shoot[i]):
            continue
        rgb = GUARD_RGB if i < num_guards else ATTACKER_RGB
        a = ang[i]
        p1 = np.array([px[i] + agent_size * np.cos(a), py[i] + agent_size * np.sin(a)])
        p2 = p1 + shoot_rad * np.array([np.cos(a + shoot_win / 2), np.sin(a + shoot_win / 2)])
        p3 = p1 + shoot_rad * np.array([np.cos(a - shoot_win / 2), np.sin(a - shoot_win / 2)])

        def side(q, r):
            return (X - q[0]) * (r[1] - q[1]) - (Y - q[1]) * (r[0] - q[0])
        s1, s2, s3 = side(p1, p2), side(p2, p3), side(p3, p1)
        tri = ((s1 >= 0) & (s2 >= 0) & (s3 >= 0)) | ((s1 <= 0) & (s2 <= 0) & (s3 <= 0))
        _blend(img, tri, rgb, 0.3)
    for i in range(n):
        if not (alive[i] or viz_dead):
            continue
        rgb = GUARD_RGB if i < num_guards else ATTACKER_RGB
        alpha = 1.0 if alive[i] else 0.35
        _blend(img, (X - px[i]) ** 2 + (Y - py[i]) ** 2 <= agent_size ** 2, rgb, alpha)
        hx, hy = px[i] + 0.8 * agent_size * np.cos(ang[i]), py[i] + 0.8 * agent_size * np.sin(ang[i])
        _blend(img, (X - hx) ** 2 + (Y - hy) ** 2 <= (0.5 * agent_size) ** 2, rgb, alpha)
    _blend(img, inside & ((Y > ymax) | (Y < ymin)), STRIP_RGB)                 # the strips of fortattack.py:548-560
    return (np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def render_state(state, env=0, num_guards=None, shoot=None, **kw):
    """One env of a `BatchedFortAttack.get_state()` dict."""
    if num_guards is None:
        raise ValueError("num_guards is required")
    return render_frame(state["pos_x"][env], state["pos_y"][env], state["ang"][env], state["alive"][env], num_guards,
                        shoot=None if shoot is None else np.asarray(shoot)[env], **kw)
