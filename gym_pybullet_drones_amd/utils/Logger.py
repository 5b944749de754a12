"""`Logger`: flight log that writes the reference's file formats (reference `utils/Logger.py`).

What callers and downstream tools see is kept: the constructor arguments, `timestamps (N, T)`, `states (N, 16, T)` in the order
`[pos3, vel3, rpy3, ang_vel3, rpm4]`, `controls (N, 12, T)`, `counters`, `log()` of one drone's 20-float state vector, `save()`
(one `.npy`-named npz archive) and `save_as_csv()` (one file per signal and drone), so files written here load wherever the
reference's do.  The storage is this package's own: one block with spare capacity that doubles when it fills, a count of the
columns in use, and the three arrays exposed as views of the used part.  `log_batch()` appends one step of ALL drones from the
`(N, 20)` block `gpd_state_vectors` produces (one device-to-host copy per logged step instead of N calls); `log()` is the
one-drone case of the same write.  Plotting is not part of this package (out of scope): load the saved
archive in the reference's `Logger` or any plotting tool.
"""
import os
from datetime import datetime

import numpy as np

# state vector (20: pos3 quat4 rpy3 vel3 ang_v3 rpm4) -> the log's 16 rows (pos3 vel3 rpy3 ang_v3 rpm4)
_LOG_ROWS = np.r_[0:3, 10:13, 7:10, 13:20]


class Logger(object):
    """In-memory flight log of NUM_DRONES drones, written out in the reference's formats."""

    def __init__(self, logging_freq_hz: int, output_folder: str = "results", num_drones: int = 1, duration_sec: int = 0,
                 colab: bool = False):
        self.COLAB = colab
        self.OUTPUT_FOLDER = output_folder
        os.makedirs(output_folder, exist_ok=True)
        self.LOGGING_FREQ_HZ = logging_freq_hz
        self.NUM_DRONES = num_drones
        self.PREALLOCATED_ARRAYS = duration_sec != 0
        self.counters = np.zeros(num_drones)
        self._used = int(duration_sec * logging_freq_hz)          # columns callers see (a pre-sized log shows all of them at once)
        cap = max(self._used, 16)
        self._t = np.zeros((num_drones, cap))
        self._x = np.zeros((num_drones, 16, cap))
        self._u = np.zeros((num_drones, 12, cap))

    timestamps = property(lambda self: self._t[:, :self._used])
    states = property(lambda self: self._x[:, :, :self._used])
    controls = property(lambda self: self._u[:, :, :self._used])

    def _column(self, cursor: int) -> int:
        """The column a write with this cursor goes to.  A pre-sized log fills its columns in order and grows by one past its
        end; a growing log always writes its newest column, opening one when the writer has caught up with it (so drones logged
        round-robin share a column per step, as in the reference)."""
        if cursor >= self._used:
            if self._used == self._t.shape[1]:
                pad = self._t.shape[1]
                self._t = np.concatenate([self._t, np.zeros((self.NUM_DRONES, pad))], axis=1)
                self._x = np.concatenate([self._x, np.zeros((self.NUM_DRONES, 16, pad))], axis=2)
                self._u = np.concatenate([self._u, np.zeros((self.NUM_DRONES, 12, pad))], axis=2)
            self._used += 1
            return self._used - 1
        return cursor if self.PREALLOCATED_ARRAYS else self._used - 1

    def _write(self, who, col, timestamp, states20, controls12):
        self._t[who, col] = timestamp
        self._x[who, :, col] = states20[..., _LOG_ROWS]
        self._u[who, :, col] = controls12
        self.counters[who] = col + 1

    def log(self, drone: int, timestamp, state, control=np.zeros(12)):
        """One step of one drone; `state` is the (20,) vector of `_getDroneStateVector`, `control` 12 targets."""
        if not (0 <= drone < self.NUM_DRONES) or timestamp < 0 or len(state) != 20 or len(control) != 12:
            print("[ERROR] in Logger.log(), invalid data")
        self._write(drone, self._column(int(self.counters[drone])), timestamp, np.asarray(state, dtype=np.float64),
                    np.asarray(control, dtype=np.float64))

    def log_batch(self, timestamp, states, controls=None):
        """One step of every drone: `states` (NUM_DRONES, 20) array or tensor (e.g. `VectorAviary.state_vectors()`
        reshaped), `controls` (NUM_DRONES, 12) or None."""
        def host(a, width):
            if hasattr(a, "detach"):
                a = a.detach().reshape(self.NUM_DRONES, width).cpu().numpy()
            return np.asarray(a, dtype=np.float64).reshape(self.NUM_DRONES, width)
        self._write(slice(None), self._column(int(self.counters.max())), timestamp, host(states, 20),
                    0.0 if controls is None else host(controls, 12))

    def trim(self):
        """Kept for callers of earlier versions: the exposed arrays never carry unused columns."""

    def _stamp(self):
        return datetime.now().strftime("%m.%d.%Y_%H.%M.%S")

    def save(self):
        """One archive `save-flight-<date>.npy` (an npz under the reference's file name) with the three arrays."""
        path = os.path.join(self.OUTPUT_FOLDER, f"save-flight-{self._stamp()}.npy")
        with open(path, "wb") as fh:
            np.savez(fh, timestamps=self.timestamps, states=self.states, controls=self.controls)
        return path

    def save_as_csv(self, comment: str = ""):
        """Save the logs as comma separated values, one file per signal and drone (reference file names)."""
        csv_dir = os.path.join(self.OUTPUT_FOLDER, "save-flight-" + comment + "-" + self._stamp())
        if not os.path.exists(csv_dir):
            os.makedirs(csv_dir + '/')
        T = self.timestamps.shape[1]
        t = np.arange(0, T / self.LOGGING_FREQ_HZ, 1 / self.LOGGING_FREQ_HZ)[:T]
        f = self.LOGGING_FREQ_HZ

        def dump(name, i, y):
            with open(csv_dir + "/" + name + str(i) + ".csv", 'wb') as out_file:
                np.savetxt(out_file, np.transpose(np.vstack([t, y])), delimiter=",")

        for i in range(self.NUM_DRONES):
            s = self.states[i]
            for name, row in (("x", 0), ("y", 1), ("z", 2), ("r", 6), ("p", 7), ("ya", 8)):
                dump(name, i, s[row])
            for name, row in (("rr", 6), ("pr", 7), ("yar", 8)):
                dump(name, i, np.hstack([0, (s[row, 1:] - s[row, 0:-1]) * f]))
            for name, row in (("vx", 3), ("vy", 4), ("vz", 5), ("wx", 9), ("wy", 10), ("wz", 11)):
                dump(name, i, s[row])
            for k in range(4):
                dump(f"rpm{k}-", i, s[12 + k])
            for k in range(4):
                dump(f"pwm{k}-", i, (s[12 + k] - 4070.3) / 0.2685)
        return csv_dir
