"""gym_collision_avoidance_amd -- MI355X-native batched collision-avoidance simulator.

Drop-in for the hot path of mit-acl/gym-collision-avoidance (`CollisionAvoidanceEnv.step` and the
Agent / Policy / Dynamics / Sensor plugin surface), with the per-step work in hand-written HIP kernels
(csrc/cagpu.hip) behind the C ABI of include/cagpu.h.  (The directory is spelled with underscores because a
Python package name cannot contain '-'.)
"""
__version__ = "0.1.0"
