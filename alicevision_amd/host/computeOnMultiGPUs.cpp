// computeOnMultiGPUs.cpp — in-process multi-GPU form: one host thread per device, each computing a contiguous chunk of the
// camera list (depthMap/computeOnMultiGPUs.cpp:15-69).  The multi-process form (one rank per GPU, pyramids exchanged over
// RCCL) is alicevision_amd/sharding.py + the CLI's --rangeStart/--rangeSize, which is also how Meshroom chunks the node.
#include "DepthMapEstimator.hpp"

#include "log.hpp"

#include <avdm.h>
#include <omp.h>

#include <algorithm>
#include <exception>

namespace avdm_host {

void computeOnMultiGPUs(const std::vector<int>& cams, IGPUJob& gpujob, int nbGPUsToUse)
{
    const int nbGPUDevices = avdm_device_count();
    const int nbCPUThreads = omp_get_max_threads();
    AVDM_LOG_INFO("Number of GPU devices: " << nbGPUDevices << ", number of CPU threads: " << nbCPUThreads);

    int nbThreads = std::min(nbGPUDevices, nbCPUThreads);
    if(nbGPUsToUse > 0)
        nbThreads = std::min(nbThreads, nbGPUsToUse);
    if(nbThreads < 1)
        throw std::runtime_error("No GPU device available.");

    if(nbThreads == 1)
    {
        gpujob.compute(0, cams);
        return;
    }
    std::exception_ptr error;
    const int previous = omp_get_max_threads();
    omp_set_num_threads(nbThreads);
#pragma omp parallel
    {
        const int cpuThreadId = omp_get_thread_num();
        const int deviceId = cpuThreadId % nbThreads;
        AVDM_LOG_INFO("CPU thread " << cpuThreadId << " (of " << nbThreads << ") uses device: " << deviceId);
        const int nbCamsPerThread = (int)(cams.size() / nbThreads);
        const int rcFrom = deviceId * nbCamsPerThread;
        int rcTo = (deviceId + 1) * nbCamsPerThread;
        if(deviceId == nbThreads - 1)
            rcTo = (int)cams.size();
        std::vector<int> subcams;
        for(int rc = rcFrom; rc < rcTo; ++rc)
            subcams.push_back(cams[rc]);
        try
        {
            if(!subcams.empty())
            {
                // the device thread runs its own (nested) parallel regions single-threaded unless nesting is enabled
                gpujob.compute(deviceId, subcams);
            }
        }
        catch(...)
        {
#pragma omp critical
            error = std::current_exception();
        }
    }
    omp_set_num_threads(previous);
    if(error)
        std::rethrow_exception(error);
}

} // namespace avdm_host
