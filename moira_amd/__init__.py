"""moira_amd -- MI355X-native Poisson-binomial read-quality filter (drop-in for moira's hot path).

Only what the path needs:
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/moira_pb.h)
  _lib.py      ctypes binding (fails loudly without the built library)
  engine.py    host-side mirror of the reference's filter interface, batch-shaped
  dropin/      `bernoulli` module with moira's calculate_errors_PB signature
  shard.py     host-side split/gather across the GPUs of a node
"""
__version__ = "0.6.0"
