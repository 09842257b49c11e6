"""CPU oracle for the dirt rasterise hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under dirt_amd/ does (tests/test_boundary.py checks that)."""
from .oracle import (build, forward, backward, visibility, draw_gl, rasterise_batch, rasterise_grad,  # noqa: F401
                     num_threads, set_num_threads, FLAG_Q1_INTENDED, FLAG_F32_SEQUENTIAL)
