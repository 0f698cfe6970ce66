"""Target-file loader: restatement of `load_targets` (/root/reference/main.py:19-32).

File format (flight_targets.txt): one header line, then `x y deter priority dx dy` per target, whitespace
separated, coordinates in units of map_size/10; deter is 't' (fixed) or 'f' (gaussian-jittered at reset).
Returns the same dict of six lists the reference builds.
"""
import os

DEFAULT_TARGETS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flight_targets.txt")


def load_targets(filename=DEFAULT_TARGETS_FILE):
    x, y, deter, priority, dx, dy = [], [], [], [], [], []
    with open(filename, "r") as f:
        rows = f.readlines()[1:]          # first line is a header
    for row in rows:
        cols = row.split()
        if not cols:
            continue
        x.append(float(cols[0]))
        y.append(float(cols[1]))
        deter.append(cols[2])
        priority.append(int(cols[3]))
        dx.append(float(cols[4]))
        dy.append(float(cols[5]))
    return {"x": x, "y": y, "deter": deter, "priority": priority, "dx": dx, "dy": dy}


def default_circle_dict():
    return load_targets(DEFAULT_TARGETS_FILE)
