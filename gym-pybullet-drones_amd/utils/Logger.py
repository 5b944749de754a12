"""`Logger`: in-memory flight log with the reference's array layout and file formats
(reference `utils/Logger.py:9-379`).

Same storage as the reference -- `timestamps (N, T)`, `states (N, 16, T)` in the re-ordered layout
`[pos3, vel3, rpy3, ang_vel3, rpm4]` (`:117`), `controls (N, 12, T)` -- the same `log()` /
`save()` / `save_as_csv()` semantics, so files written here load wherever the reference's do.
New: `log_batch()` appends one step of ALL drones from the `(N, 20)` state-vector block the
engine's `gpd_state_vectors` produces (one device-to-host copy per logged step instead of N calls).
`plot()` needs matplotlib (absent in this image) and raises if it is missing.
"""
import os
from datetime import datetime

import numpy as np


class Logger(object):
    """A class for logging and visualization."""

    def __init__(self, logging_freq_hz: int, output_folder: str = "results", num_drones: int = 1, duration_sec: int = 0,
                 colab: bool = False):
        self.COLAB = colab
        self.OUTPUT_FOLDER = output_folder
        if not os.path.exists(self.OUTPUT_FOLDER):
            os.mkdir(self.OUTPUT_FOLDER)
        self.LOGGING_FREQ_HZ = logging_freq_hz
        self.NUM_DRONES = num_drones
        self.PREALLOCATED_ARRAYS = False if duration_sec == 0 else True
        self.counters = np.zeros(num_drones)
        self.timestamps = np.zeros((num_drones, duration_sec * self.LOGGING_FREQ_HZ))
        #### 16 states: pos_x, pos_y, pos_z, vel_x, vel_y, vel_z, roll, pitch, yaw, ang_vel_x, ang_vel_y, ang_vel_z, rpm0-3
        self.states = np.zeros((num_drones, 16, duration_sec * self.LOGGING_FREQ_HZ))
        #### 12 control targets: pos, vel, rpy, ang_vel
        self.controls = np.zeros((num_drones, 12, duration_sec * self.LOGGING_FREQ_HZ))

    ################################################################################

    def log(self, drone: int, timestamp, state, control=np.zeros(12)):
        """Logs one step of one drone; `state` is the (20,) vector of `_getDroneStateVector`."""
        if drone < 0 or drone >= self.NUM_DRONES or timestamp < 0 or len(state) != 20 or len(control) != 12:
            print("[ERROR] in Logger.log(), invalid data")
        current_counter = int(self.counters[drone])
        #### Add rows to the matrices if a counter exceeds their size
        if current_counter >= self.timestamps.shape[1]:
            self.timestamps = np.concatenate((self.timestamps, np.zeros((self.NUM_DRONES, 1))), axis=1)
            self.states = np.concatenate((self.states, np.zeros((self.NUM_DRONES, 16, 1))), axis=2)
            self.controls = np.concatenate((self.controls, np.zeros((self.NUM_DRONES, 12, 1))), axis=2)
        #### Advance a counter is the matrices have overgrown it ###
        elif not self.PREALLOCATED_ARRAYS and self.timestamps.shape[1] > current_counter:
            current_counter = self.timestamps.shape[1] - 1
        self.timestamps[drone, current_counter] = timestamp
        #### Re-order the kinematic obs (of most Aviaries) #########
        self.states[drone, :, current_counter] = np.hstack([state[0:3], state[10:13], state[7:10], state[13:20]])
        self.controls[drone, :, current_counter] = control
        self.counters[drone] = current_counter + 1

    def log_batch(self, timestamp, states, controls=None):
        """Logs one step of every drone: `states` (NUM_DRONES, 20) array or tensor (e.g. `VectorAviary.state_vectors()`
        reshaped), `controls` (NUM_DRONES, 12) or None."""
        if hasattr(states, "detach"):
            states = states.detach().reshape(self.NUM_DRONES, 20).cpu().numpy()
        states = np.asarray(states, dtype=np.float64).reshape(self.NUM_DRONES, 20)
        if controls is None:
            controls = np.zeros((self.NUM_DRONES, 12))
        elif hasattr(controls, "detach"):
            controls = controls.detach().reshape(self.NUM_DRONES, 12).cpu().numpy()
        c = int(self.counters.max())
        if c >= self.timestamps.shape[1]:
            grow = max(1, self.timestamps.shape[1])            # amortised doubling instead of one column per call
            self.timestamps = np.concatenate((self.timestamps, np.zeros((self.NUM_DRONES, grow))), axis=1)
            self.states = np.concatenate((self.states, np.zeros((self.NUM_DRONES, 16, grow))), axis=2)
            self.controls = np.concatenate((self.controls, np.zeros((self.NUM_DRONES, 12, grow))), axis=2)
            self._slack = True
        self.timestamps[:, c] = timestamp
        self.states[:, :, c] = np.hstack([states[:, 0:3], states[:, 10:13], states[:, 7:10], states[:, 13:20]])
        self.controls[:, :, c] = controls
        self.counters[:] = c + 1

    def trim(self):
        """Drop the unused columns `log_batch` may have pre-grown."""
        c = int(self.counters.max())
        self.timestamps, self.states, self.controls = self.timestamps[:, :c], self.states[:, :, :c], self.controls[:, :, :c]

    ################################################################################

    def save(self):
        """Save the logs to file."""
        if getattr(self, "_slack", False):
            self.trim()
        with open(os.path.join(self.OUTPUT_FOLDER, "save-flight-" + datetime.now().strftime("%m.%d.%Y_%H.%M.%S") + ".npy"),
                  'wb') as out_file:
            np.savez(out_file, timestamps=self.timestamps, states=self.states, controls=self.controls)

    def save_as_csv(self, comment: str = ""):
        """Save the logs as comma separated values, one file per signal and drone (reference file names)."""
        if getattr(self, "_slack", False):
            self.trim()
        csv_dir = os.path.join(self.OUTPUT_FOLDER, "save-flight-" + comment + "-" + datetime.now().strftime("%m.%d.%Y_%H.%M.%S"))
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

    def plot(self, pwm=False):
        """10x2 grid of position/velocity/attitude/rate/RPM traces (reference `:205-379`); needs matplotlib."""
        try:
            import matplotlib.pyplot as plt
        except ImportError as e:                                # pragma: no cover - matplotlib is not in this image
            raise ImportError("Logger.plot() needs matplotlib") from e
        if getattr(self, "_slack", False):
            self.trim()
        T = self.timestamps.shape[1]
        t = np.arange(0, T / self.LOGGING_FREQ_HZ, 1 / self.LOGGING_FREQ_HZ)[:T]
        fig, axs = plt.subplots(10, 2)
        left = [(0, 'x (m)'), (1, 'y (m)'), (2, 'z (m)'), (6, 'r (rad)'), (7, 'p (rad)'), (8, 'y (rad)'),
                (9, 'wx'), (10, 'wy'), (11, 'wz')]
        right = [(3, 'vx (m/s)'), (4, 'vy (m/s)'), (5, 'vz (m/s)')]
        for row, (idx, label) in enumerate(left):
            for j in range(self.NUM_DRONES):
                axs[row, 0].plot(t, self.states[j, idx, :], label="drone_" + str(j))
            axs[row, 0].set_ylabel(label)
        for row, (idx, label) in enumerate(right):
            for j in range(self.NUM_DRONES):
                axs[row, 1].plot(t, self.states[j, idx, :], label="drone_" + str(j))
            axs[row, 1].set_ylabel(label)
        for row, idx in ((3, 6), (4, 7), (5, 8)):
            for j in range(self.NUM_DRONES):
                d = np.hstack([0, (self.states[j, idx, 1:] - self.states[j, idx, 0:-1]) * self.LOGGING_FREQ_HZ])
                axs[row, 1].plot(t, d, label="drone_" + str(j))
        for k in range(4):
            for j in range(self.NUM_DRONES):
                y = self.states[j, 12 + k, :]
                axs[6 + k, 1].plot(t, (y - 4070.3) / 0.2685 if pwm else y, label="drone_" + str(j))
            axs[6 + k, 1].set_ylabel(('PWM' if pwm else 'RPM') + str(k))
        for ax in axs.flat:
            ax.grid(True)
        if not self.COLAB:
            plt.show()
        else:
            plt.savefig(os.path.join('results', 'output_figure.png'))
