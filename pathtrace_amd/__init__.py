"""pathtrace_amd -- MI355X-native backend for the hot path of 0Xero7/pathtrace.

BVH traversal (closest hit / any hit), ray-triangle intersection and the multi-bounce
integrator run as hand-written HIP kernels in libptmi.so behind the C ABI of include/ptmi.h;
this package is the thin host-side mirror of the reference's Go interface for that path.
"""
from .api import (BuildBVH, ConstructLinearBVH, UploadLinearBVH, LinearBVH, Box, Context, LinearArrays,
                  PtmiError)

__all__ = ["BuildBVH", "ConstructLinearBVH", "UploadLinearBVH", "LinearBVH", "Box", "Context", "LinearArrays",
           "PtmiError"]
