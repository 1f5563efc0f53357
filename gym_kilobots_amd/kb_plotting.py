"""Debug view of ONE env of a batched simulation: matplotlib drawing of the arena, the
kilobots (disc + heading), circular objects and circular lights, read from the device state.  This replaces
the role of the reference's pygame viewer (gym_kilobots/kb_rendering.py) and matplotlib helpers
(gym_kilobots/kb_plotting.py) for eyeballing contact behaviour; it is not on the hot path."""
import numpy as np

KILOBOT_RADIUS = 0.0165


def snapshot(sim, env_index=0):
    """Poses of env `env_index` as NumPy arrays (metres): kilobots [N,3], objects [M,3] or None, light [L,2] or None."""
    kb = sim.poses()[env_index].detach().cpu().numpy()
    objs = sim.object_poses()[env_index].detach().cpu().numpy() if getattr(sim, 'num_objects', 0) else None
    light = None
    if getattr(sim, 'light_x', None) is not None and getattr(sim, 'light_y', None) is not None:
        lx = np.atleast_1d(sim.light_x[env_index].detach().cpu().numpy())
        ly = np.atleast_1d(sim.light_y[env_index].detach().cpu().numpy())
        light = np.stack([lx, ly], -1)
    return kb, objs, light


def plot_body(axes, body, **kwargs):
    """Body.plot of the reference (kb_plotting.py:44-145 plot_rect / plot_polygon / plot_circle): one object view."""
    from matplotlib.patches import Circle as MplCircle, Polygon as MplPolygon
    kwargs.setdefault('facecolor', tuple(np.asarray(body.color) / 255.0))
    kwargs.setdefault('edgecolor', 'k')
    if hasattr(body, 'get_radius'):
        return axes.add_patch(MplCircle(tuple(body.get_position()), body.get_radius(), **kwargs))
    return [axes.add_patch(MplPolygon(np.asarray(vs), closed=True, **kwargs)) for vs in body.vertices]


def plot_env(sim, env_index=0, axes=None, world_size=(2.0, 1.5), object_radii=None, light_radii=None, title=None,
             object_vertices=None):
    """Draw env `env_index` of `sim` (a KilobotSim or anything with the same poses()/object_poses() API).
    object_vertices: per object None (circle of object_radii) or its body-frame polygon [[x, y], ...] in metres."""
    import matplotlib
    if axes is None:
        matplotlib.use('Agg', force=False)
    import matplotlib.pyplot as plt
    from matplotlib.patches import Circle as MplCircle, Rectangle
    kb, objs, light = snapshot(sim, env_index)
    if axes is None:
        _, axes = plt.subplots(figsize=(8, 8 * world_size[1] / world_size[0]))
    w, h = world_size
    axes.add_patch(Rectangle((-w / 2, -h / 2), w, h, fill=False, linewidth=1.5, edgecolor='k'))
    if light is not None:
        radii = light_radii if light_radii is not None else [0.2] * len(light)
        for (lx, ly), r in zip(light, radii):
            axes.add_patch(MplCircle((lx, ly), r, facecolor=(1.0, 1.0, 0.12, 0.35), edgecolor='none'))
    if objs is not None:
        radii = object_radii if object_radii is not None else [0.075] * len(objs)
        polys = object_vertices if object_vertices is not None else [None] * len(objs)
        for (x, y, th), r, vs in zip(objs, radii, polys):
            if vs is not None:
                from matplotlib.patches import Polygon as MplPolygon
                c, s_ = np.cos(th), np.sin(th)
                vw = [(c * vx - s_ * vy + x, s_ * vx + c * vy + y) for vx, vy in vs]
                axes.add_patch(MplPolygon(vw, closed=True, facecolor=(93 / 255, 133 / 255, 195 / 255), edgecolor='k', linewidth=0.5))
                continue
            axes.add_patch(MplCircle((x, y), r, facecolor=(93 / 255, 133 / 255, 195 / 255), edgecolor='k', linewidth=0.5))
            axes.plot([x, x + r * np.cos(th)], [y, y + r * np.sin(th)], color='k', linewidth=0.5)
    for x, y, th in kb:
        axes.add_patch(MplCircle((x, y), KILOBOT_RADIUS, facecolor=(0.6, 0.6, 0.6), edgecolor=(0.4, 0.4, 0.4), linewidth=0.5))
        axes.plot([x, x + KILOBOT_RADIUS * np.cos(th)], [y, y + KILOBOT_RADIUS * np.sin(th)], color='w', linewidth=0.8)
    axes.set_xlim(-w / 2 - 0.02, w / 2 + 0.02)
    axes.set_ylim(-h / 2 - 0.02, h / 2 + 0.02)
    axes.set_aspect('equal')
    if title:
        axes.set_title(title)
    return axes


def save_env_png(sim, path, env_index=0, **kw):
    import matplotlib
    matplotlib.use('Agg', force=False)
    import matplotlib.pyplot as plt
    ax = plot_env(sim, env_index, **kw)
    ax.figure.savefig(path, dpi=120, bbox_inches='tight')
    plt.close(ax.figure)
    return path
