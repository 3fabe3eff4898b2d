"""Key/value training log with the interface ``mjrl.utils.logger.DataLog`` exposes to agents
and ``train_agent`` (reference mjrl/utils/logger.py:10-81): log_kv / save_log /
get_current_log / shrink_to / read_log.  mjrl's own DataLog is the base class when mjrl is importable.

One addition: entries may be logged BEFORE their value exists (``PendingValue``: the errors and the duration of a baseline fit
that is still running on a side stream, algos/batch_reinforce.py).  Reading such an entry as a number waits for it; ``save_log``
settles every pending entry first, so log.csv / log.pickle only ever hold numbers; str() of one still in flight is "(fitting)".
"""
import csv
import os
import pickle

try:                                                    # pragma: no cover - depends on the host env
    from mjrl.utils.logger import DataLog as _Base
except Exception:
    class _Base:
        def __init__(self):
            self.log = {}
            self.max_len = 0

        def log_kv(self, key, value):
            series = self.log.setdefault(key, [])
            series.append(value)
            self.max_len = max(self.max_len, len(series))

        def get_current_log(self):
            return {k: v[-1] for k, v in self.log.items() if len(v)}

        def shrink_to(self, num_entries):
            for k in self.log:
                self.log[k] = self.log[k][:num_entries]
            self.max_len = num_entries

        def save_log(self, save_path):
            os.makedirs(save_path, exist_ok=True)
            with open(os.path.join(save_path, "log.pickle"), "wb") as f:
                pickle.dump(self.log, f)
            keys = sorted(self.log.keys())
            with open(os.path.join(save_path, "log.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["iteration"] + [k for k in keys if k != "iteration"])
                for i in range(self.max_len):
                    w.writerow([i] + [self.log[k][i] if i < len(self.log[k]) else "" for k in keys if k != "iteration"])

        def read_log(self, log_path):
            with open(log_path) as f:
                rows = list(csv.DictReader(f))
            keys = [k for k in rows[0].keys() if k != "iteration"] if rows else []
            self.log = {k: [] for k in keys}
            for row in rows:
                for k in keys:
                    if row[k] != "":
                        try:
                            self.log[k].append(float(row[k]))
                        except ValueError:
                            pass
            self.max_len = max([len(v) for v in self.log.values()] + [0])


class PendingValue:
    """A log entry whose value arrives later.  ``source.result()`` must make it arrive (it calls ``deliver``)."""
    __slots__ = ("source", "value", "ready")

    def __init__(self, source):
        self.source, self.value, self.ready = source, None, False

    def deliver(self, value):
        self.value, self.ready, self.source = float(value), True, None

    def get(self):
        if not self.ready:
            self.source.result()                         # waits; its hooks deliver
        return self.value

    def poll(self):
        """-> True when the number is there (takes it over if the work behind it has finished; never waits)"""
        if not self.ready and self.source is not None and getattr(self.source, "finished", lambda: False)():
            self.get()
        return self.ready

    def __float__(self):
        return self.get()

    def __reduce__(self):                                # a pickle of the log holds the number
        return (float, (self.get(),))

    def __str__(self):
        return repr(self.value) if self.poll() else "(fitting)"

    __repr__ = __str__

    def __format__(self, spec):
        return format(self.value, spec) if self.ready else "(fitting)"

    # arithmetic / comparisons read the number (make_train_plots.py:30 scales log entries)
    def __mul__(self, o): return self.get() * o
    def __rmul__(self, o): return o * self.get()
    def __add__(self, o): return self.get() + o
    def __radd__(self, o): return o + self.get()
    def __sub__(self, o): return self.get() - o
    def __rsub__(self, o): return o - self.get()
    def __truediv__(self, o): return self.get() / o
    def __rtruediv__(self, o): return o / self.get()
    def __neg__(self): return -self.get()
    def __lt__(self, o): return self.get() < o
    def __le__(self, o): return self.get() <= o
    def __gt__(self, o): return self.get() > o
    def __ge__(self, o): return self.get() >= o
    def __eq__(self, o): return self.get() == o
    __hash__ = None


class DataLog(_Base):
    def settle(self):
        """replace every pending entry by its number (waits for the work behind it)"""
        for series in self.log.values():
            for i, v in enumerate(series):
                if isinstance(v, PendingValue):
                    series[i] = v.get()

    def save_log(self, save_path):
        self.settle()
        return super().save_log(save_path)

    def get_current_log(self):
        """the latest row; entries whose work has finished meanwhile are numbers, one still in flight stays a PendingValue
        (float() of it waits).  NOTE: the reference's train_agent prints this row through tabulate after every iteration
        (train_agent.py:150-153) when save_logs is set, and tabulate probes every cell with float(): under that driver the fit is
        waited for at the print -- correct numbers on the console, no overlap with the next rollouts.  Loops that read the log
        less often (or not at all) keep the overlap."""
        row = super().get_current_log()
        return {k: (v.get() if isinstance(v, PendingValue) and v.poll() else v) for k, v in row.items()}
