"""gym_kilobots_amd: MI355X-native batched Kilobot swarm simulation (drop-in for the hot path of
gregorgebhardt/gym-kilobots).  `envs` and `lib` mirror the reference's namespaces; `sim.KilobotSim`
is the device-side simulator; `_native` is the ctypes binding of libkilobots_hip.so."""
from . import _native  # noqa: F401

__all__ = ['envs', 'lib', 'sim', 'dist', 'spaces']
