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


def gantt(env):
    """Plotly figure of ``env.solution`` or None when nothing is scheduled yet."""
    if not hasattr(env, "_render_t0"):
        env._render_t0 = datetime.datetime.now().timestamp()
        env._render_colors = [tuple(random.random() for _ in range(3)) for _ in range(env.machines)]
    rows = gantt_rows(env.solution, env.instance, env._render_t0)
    if not rows:
        return None
    import pandas as pd
    import plotly.figure_factory as ff
    fig = ff.create_gantt(pd.DataFrame(rows), index_col="Resource", colors=env._render_colors, show_colorbar=True,
                          group_tasks=True)
    fig.update_yaxes(autorange="reversed")   # tasks listed top-down
    return fig
