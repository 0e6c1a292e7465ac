"""Builder-owned FUNCTIONAL fake of the slice of pycuda that instaGRAAL's sampler uses.

Purpose: let the reference's own, unmodified
``instagraal.cuda_lib_gl_single.sampler`` Python run in the authoring container
(no CUDA, no nvcc) so that its host orchestration -- candidate draw, stale
flags, argmax, apply, RNG consumption -- can be captured as golden vectors.
"Device memory" is host numpy memory; ``SourceModule.get_function(name)``
returns the CPU restatement of that kernel from oracle/ig_oracle_*.c, taking
the reference's exact argument list.  Used only by tools/gen_golden.py.
"""
VERSION = (0, 0, 0)
