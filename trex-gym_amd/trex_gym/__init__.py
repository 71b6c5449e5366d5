"""MI355X-native drop-in for the hot path of bingjeff/trex-gym: the batched physics step behind
trex_gym.trex_env.TrexBulletEnv. Import path kept: `trex_gym.trex_env.TrexBulletEnv`."""
