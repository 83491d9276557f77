"""Host-only Gantt view of one env's ``solution`` (reference: JssEnv.render, jss_env.py:655-693).

Not on the metric's path: it pulls ``solution`` (start times, -1 = unscheduled) of ONE env to
the host and hands the bars to plotly's ``create_gantt`` like the reference does.  pandas and
plotly are imported lazily so the package has no hard dependency on them.
"""
from __future__ import annotations

import datetime
import random


def gantt_rows(solution, instance, start_timestamp: float):
    """One dict per scheduled op (Task / Start / Finish / Resource), in job-major order."""
    rows = []
    for job in range(instance.jobs):
        for k in range(instance.machines):
            start = int(solution[job][k])
            if start == -1:          # ops are scheduled in order: the rest of the job is unscheduled
                break
            begin = start_timestamp + start
            rows.append({
                "Task": f"Job {job}",
                "Start": datetime.datetime.fromtimestamp(begin),
                "Finish": datetime.datetime.fromtimestamp(begin + int(instance.duration[job][k])),
                "Resource": f"Machine {int(instance.machine[job][k])}",
            })
    return rows


def _ensure_render_attrs(env):
    """``start_timestamp`` (jss_env.py:70) and ``colors`` (:99-101) are public attributes of the reference's env, set by its
    constructor; JssEnv sets them too -- any other object handed to the helpers here gets them on first use."""
    if not hasattr(env, "start_timestamp"):
        env.start_timestamp = datetime.datetime.now().timestamp()
    if not hasattr(env, "colors"):
        env.colors = [tuple(random.random() for _ in range(3)) for _ in range(env.machines)]


def gantt(env):
    """Plotly figure of ``env.solution`` or None when nothing is scheduled yet."""
    _ensure_render_attrs(env)
    rows = gantt_rows(env.solution, env.instance, env.start_timestamp)
    if not rows:
        return None
    import pandas as pd
    import plotly.figure_factory as ff
    fig = ff.create_gantt(pd.DataFrame(rows), index_col="Resource", colors=env.colors, show_colorbar=True,
                          group_tasks=True)
    fig.update_yaxes(autorange="reversed")   # tasks listed top-down
    return fig


# ---------------------------------------------------------------------------------------------------------
# Animated GIFs of a schedule growing step by step (reference recipe: README.md:158-197, tests/test_rendering.py:65-79:
# ``imageio.imread(env.render().to_image())`` per step, ``imageio.mimsave`` at the end).  Both third-party pieces are
# optional here: a frame is plotly's own raster when plotly can make one (kaleido installed), otherwise the same bars
# drawn with Pillow; the file is written by imageio when it is importable, otherwise by Pillow.
# ---------------------------------------------------------------------------------------------------------
def _machine_colors(env):
    """One RGB triple (0-255) per machine: the colours render() hands to plotly (which rewrites that list in place as
    'rgb(r, g, b)' strings -- both spellings are read here)."""
    _ensure_render_attrs(env)
    out = []
    for c in env.colors:
        if isinstance(c, str):
            out.append(tuple(int(float(x)) for x in c[c.index("(") + 1:c.index(")")].split(",")))
        else:
            out.append(tuple(int(255 * x) for x in c))
    return out


def gantt_frame(env, size=(960, 540), horizon=None, prefer_plotly=True):
    """The current schedule of ``env`` as an (H, W, 3) uint8 image, or None when nothing is scheduled yet.

    ``horizon``: time span of the x axis (default: the instance's total work / machines * 2, clipped to what is
    scheduled) -- pass the same value for every frame of an animation so the bars do not rescale."""
    import numpy as np
    rows = gantt_rows(env.solution, env.instance, 0.0)
    if not rows:
        return None
    if prefer_plotly:
        try:                                   # the reference's way: plotly rasterises its own figure (needs kaleido)
            import io
            from PIL import Image
            png = env.render().to_image(format="png", width=size[0], height=size[1])
            return np.asarray(Image.open(io.BytesIO(png)).convert("RGB"))
        except Exception:
            pass
    from PIL import Image, ImageDraw
    W, H = size
    colors = _machine_colors(env)
    epoch = datetime.datetime.fromtimestamp(0.0)
    bars = [(int(r["Task"].split()[1]), int(r["Resource"].split()[1]), (r["Start"] - epoch).total_seconds(),
             (r["Finish"] - epoch).total_seconds()) for r in rows]
    span = max(b[3] for b in bars)
    if horizon is not None:
        span = max(span, float(horizon))
    left, top, right, bottom = 70, 20, W - 20, H - 30
    lane = (bottom - top) / env.jobs
    img = Image.new("RGB", (W, H), (255, 255, 255))
    d = ImageDraw.Draw(img)
    for job in range(env.jobs):                 # tasks listed top-down, like the reference's reversed y axis
        y0 = top + job * lane
        d.text((5, y0 + lane * 0.2), f"Job {job}", fill=(0, 0, 0))
        d.line([(left, y0), (right, y0)], fill=(230, 230, 230))
    for job, m, t0, t1 in bars:
        x0 = left + (right - left) * t0 / span
        x1 = left + (right - left) * t1 / span
        y0 = top + job * lane + lane * 0.15
        y1 = top + (job + 1) * lane - lane * 0.15
        d.rectangle([x0, y0, max(x1, x0 + 1), y1], fill=colors[m], outline=(40, 40, 40))
    d.line([(left, bottom), (right, bottom)], fill=(0, 0, 0))
    d.text((left, bottom + 8), "0", fill=(0, 0, 0))
    d.text((right - 40, bottom + 8), str(int(span)), fill=(0, 0, 0))
    return np.asarray(img)


def save_gif(frames, path, fps: float = 10.0):
    """Write ``frames`` ((H, W, 3) uint8 arrays) as an animated GIF at ``fps`` frames per second: through Pillow (which is
    also what imageio's GIF plug-in writes with), imageio itself only where Pillow is missing.  Returns the number of frames."""
    frames = [f for f in frames if f is not None]
    if not frames:
        raise ValueError("no frames to write (nothing was scheduled)")
    try:
        from PIL import Image          # one writer, one unit: Pillow's duration is milliseconds per frame
    except ImportError:
        # imageio's own GIF writer: `duration` is seconds per frame up to 2.27 and MILLISECONDS from 2.28 on (the v3 pillow
        # plug-in) -- the same literal would play a thousand times too fast on one side of that line
        import imageio
        new_units = tuple(int(x) for x in imageio.__version__.split(".")[:2] if x.isdigit()) >= (2, 28)
        imageio.mimsave(str(path), frames, duration=(1000.0 if new_units else 1.0) / fps)
        return len(frames)
    imgs = [Image.fromarray(f) for f in frames]
    imgs[0].save(str(path), save_all=True, append_images=imgs[1:], duration=int(round(1000.0 / fps)), loop=0)
    return len(frames)


def record_episode_gif(env, choose_action, path, every: int = 1, fps: float = 10.0, size=(960, 540), max_steps=None):
    """The README recipe as one call: ``reset()``, then ``obs, r, done, _, _ = env.step(choose_action(env))`` until the
    episode ends, one frame per ``every`` steps (and the final schedule), written to ``path``.  Works on either
    backend (the env's ``solution`` is pulled from the device per captured frame).  Returns (frames, makespan)."""
    env.reset()
    horizon = None
    frames, done, n = [], False, 0
    while not done and (max_steps is None or n < max_steps):
        _, _, done, _, _ = env.step(choose_action(env))
        n += 1
        if n % every == 0 or done:
            if horizon is None:
                horizon = 2.0 * float(env.sum_op) / env.machines
            frames.append(gantt_frame(env, size=size, horizon=horizon))
    return save_gif(frames, path, fps=fps), env.current_time_step
