"""Key/value training log with the interface ``mjrl.utils.logger.DataLog`` exposes to agents
and ``train_agent`` (reference mjrl/utils/logger.py:10-81): log_kv / save_log /
get_current_log / shrink_to / read_log.  mjrl's own DataLog is used when mjrl is importable."""
import csv
import os
import pickle

try:                                                    # pragma: no cover - depends on the host env
    from mjrl.utils.logger import DataLog               # noqa: F401
except Exception:
    class DataLog:
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
                w.writerow(["iteration"] + keys)
                for i in range(self.max_len):
                    w.writerow([i] + [self.log[k][i] if i < len(self.log[k]) else "" for k in keys])

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
