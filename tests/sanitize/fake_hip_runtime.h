// interface of the fake HIP runtime (fake_hip_runtime.cpp) towards a test harness: models of kernels
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>

namespace fake_hip {
// called instead of the kernel whose (mangled) name contains the key: args[i] points to the i-th launch argument
using Model = std::function<void(void** args, dim3 grid, dim3 block)>;
void set_model(const std::string& kernel_substring, Model m);
void clear_models();
long launches();
long allocations();
void fail_allocation_in(int n); // the n-th hipMalloc from now on fails once
}  // namespace fake_hip
