"""CPU oracle for the PPO hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  Nothing under ``ppo_amd/`` does.

* ``oracle.cref``  -- ctypes binding of ``ppo_oracle.c`` (scalar C restatement).
* ``oracle.pyref`` -- per-env Python objects + the reference's serial ``SyncVecEnv`` /
  GAE loops restated (what the reference's CPU path does, one core).
"""
