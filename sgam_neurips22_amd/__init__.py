"""sgam_neurips22_amd — MI355X (gfx950) native backend of SGAM's per-step generative-sensing hot path.

Layout:
  csrc/ + lib/libsgam_hip.so   hand-written HIP kernels behind the C ABI of include/sgam_hip.h
  _lib.py, ops.py              ctypes binding + tensor-level wrappers (torch = device memory/streams only)
  generative_sensing_module/   VQModel / Encoder / Decoder / VectorQuantizer2 with the reference's call surface
  point_rendering/warp.py      render_projection_from_srcs_fast (forward splat)
  inference_pipeline.py        InfiniteSceneGeneration counterpart (in-memory trajectory runner, inverse warp)
"""
__version__ = "0.1.0"
