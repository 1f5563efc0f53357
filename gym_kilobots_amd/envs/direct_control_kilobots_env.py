"""DirectControlKilobotsEnv (reference gym_kilobots/envs/direct_control_kilobots_env.py): per-kilobot
actions, the "random motor commands" entry point of BASELINE config 1."""
import numpy as np
import torch

from ..spaces import Box
from .kilobots_env import KilobotsEnv


class DirectControlKilobotsEnv(KilobotsEnv):
    def __init__(self, **kwargs):
        super(DirectControlKilobotsEnv, self).__init__(**kwargs)

    @property
    def action_space(self):
        as_low = np.array([kb.action_space.low for kb in self._kilobots])
        as_high = np.array([kb.action_space.high for kb in self._kilobots])
        return Box(as_low, as_high, dtype=np.float64)

    def step(self, actions):
        if self._sim is None:
            raise RuntimeError('call reset() before step()')
        if actions is not None:
            if torch.is_tensor(actions):
                a = actions.to(device=self._sim.x.device, dtype=torch.float32)
            else:
                a = torch.as_tensor(np.asarray(actions, dtype=np.float32), device=self._sim.x.device)
            if a.dim() == 2:
                a = a.unsqueeze(0).expand(self.num_envs, -1, -1)
            # kb.set_action(a) for every kilobot (clamped on the device, kilobot.py:235-241)
            self._sim.set_actions(a.contiguous())
        else:
            self._sim.set_actions(None)
        return super(DirectControlKilobotsEnv, self).step(None)
