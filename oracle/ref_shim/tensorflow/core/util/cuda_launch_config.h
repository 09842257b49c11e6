/* Host shim (TEST INFRASTRUCTURE, see ../framework/tensor.h): TensorFlow's 2-D launch configuration,
   reduced to one block of one thread covering the whole virtual range. */
#ifndef DIRT_REF_SHIM_LAUNCH_CONFIG_H
#define DIRT_REF_SHIM_LAUNCH_CONFIG_H

#include <tensorflow/core/framework/tensor.h>

namespace tensorflow {
    struct CudaLaunchConfig2D {
        dim3 virtual_thread_count, thread_per_block, block_count;
    };
    inline CudaLaunchConfig2D GetCuda2DLaunchConfig(int xdim, int ydim, Eigen::GpuDevice const &) {
        CudaLaunchConfig2D config;
        config.virtual_thread_count = dim3(xdim, ydim, 1);
        config.thread_per_block = dim3(1, 1, 1);
        config.block_count = dim3(1, 1, 1);
        return config;
    }
}

#endif
