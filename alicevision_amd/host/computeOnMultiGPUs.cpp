// computeOnMultiGPUs.cpp — the multi-GPU driver of a camera job inside one process.
//
// Interface of the reference (depthMap/computeOnMultiGPUs.hpp:18-36: a job with compute(deviceId, cams), a camera list, the number of
// GPUs to use), re-designed behind it (BASELINE north_star):
//   * one host THREAD per device (std::thread, so that the OpenMP regions inside a job — image decoding, depth lists, tile merging —
//     keep their own teams; inside the reference's `#pragma omp parallel` they would be nested and run single-threaded);
//   * R cameras are dealt ROUND-ROBIN to the devices (the reference cuts the list into contiguous chunks, computeOnMultiGPUs.cpp:49-63;
//     depth maps are independent, so the results are the same — consecutive cameras have similar cost, dealing them balances better);
//   * jobs that implement viewsNeeded() / computeShared() share one PyramidExchange: every view of the job is decoded and converted
//     by exactly one device and reaches the others as a finished pyramid over the fabric (device.hpp).
// AVDM_FAKE_DEVICES=n runs n workers on the devices that exist (worker w on device w % count): how the one-GPU test box exercises the
// whole multi-worker path, including the peer copies.
#include "DepthMapEstimator.hpp"

#include "device.hpp"
#include "log.hpp"

#include <avdm.h>
#include <hip/hip_runtime.h>
#include <omp.h>

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>

namespace avdm_host {

void computeOnMultiGPUs(const std::vector<int>& cams, IGPUJob& gpujob, int nbGPUsToUse)
{
    const int nbPhysical = avdm_device_count();
    if(nbPhysical < 1)
        throw std::runtime_error("No GPU device available.");
    int nbWorkers = nbPhysical;
    if(const char* fake = std::getenv("AVDM_FAKE_DEVICES"))
        nbWorkers = std::max(1, std::atoi(fake));
    if(nbGPUsToUse > 0)
        nbWorkers = std::min(nbWorkers, nbGPUsToUse);
    nbWorkers = std::max(1, std::min(nbWorkers, (int)std::max<size_t>(cams.size(), 1)));
    const int hostThreads = omp_get_max_threads();
    AVDM_LOG_INFO("Number of GPU devices: " << nbPhysical << ", workers: " << nbWorkers << ", CPU threads: " << hostThreads);

    if(nbWorkers == 1)
    {
        gpujob.compute(0, cams);
        return;
    }

    std::vector<int> devices(nbWorkers);
    std::vector<std::vector<int>> share(nbWorkers);
    for(int w = 0; w < nbWorkers; ++w)
        devices[w] = w % nbPhysical;
    for(size_t i = 0; i < cams.size(); ++i)
        share[i % nbWorkers].push_back(cams[i]);

    const std::vector<int> allViews = gpujob.viewsNeeded(cams);
    PyramidExchange exchange(devices);
    // Residency budget of the exchange per owner: a quarter of the smallest device's memory (AVDM_EXCHANGE_BUDGET_MB overrides; the reference
    // keeps nbRcPerBatch * (1 + maxTCams) pyramids per device and decodes every neighbour on every device).  Views beyond it are declined
    // and decoded by whoever needs them (device.hpp).
    {
        size_t budget = (size_t)-1;
        for(int d = 0; d < nbPhysical; ++d)
        {
            size_t freeB = 0, totalB = 0;
            if(hipSetDevice(d) == hipSuccess && hipMemGetInfo(&freeB, &totalB) == hipSuccess)
                budget = std::min(budget, totalB / 4 / std::max<size_t>(1, (nbWorkers + nbPhysical - 1) / nbPhysical));
        }
        if(const char* mb = std::getenv("AVDM_EXCHANGE_BUDGET_MB"))
            budget = (size_t)std::max(0L, std::atol(mb)) << 20;
        exchange.setBudgetBytes(budget);
        AVDM_LOG_INFO("Pyramid exchange: residency budget " << (budget >> 20) << " MB per worker.");
    }
    // Peer access between every pair of devices in use, BEFORE the first hipMemcpyPeerAsync: with it the copies go device to device over
    // xGMI; without it the runtime may stage them through host memory.  (The one-process form of BASELINE's "neighbour views broadcast
    // once over RCCL/xGMI": threads of one process share an address space, so a peer copy IS the broadcast; the one-process-per-GPU form —
    // bench.py, alicevision_amd/sharding.py — uses RCCL.)
    for(int a = 0; a < std::min(nbWorkers, nbPhysical); ++a)
        for(int b = 0; b < std::min(nbWorkers, nbPhysical); ++b)
        {
            if(a == b)
                continue;
            int can = 0;
            if(hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can)
            {
                AVDM_LOG_WARNING("Device " << a << " cannot access device " << b << " directly: pyramid copies between them are staged by the runtime.");
                continue;
            }
            if(hipSetDevice(a) != hipSuccess)
                continue;
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                AVDM_LOG_WARNING("hipDeviceEnablePeerAccess(" << a << " -> " << b << ") failed: " << hipGetErrorString(e));
            else
                AVDM_LOG_DEBUG("Peer access " << a << " -> " << b << " enabled: pyramid copies are direct.");
            (void)hipGetLastError();
        }
    (void)hipSetDevice(0);

    std::mutex errorGuard;
    std::exception_ptr firstError;
    std::vector<std::thread> threads;
    for(int w = 0; w < nbWorkers; ++w)
        threads.emplace_back([&, w] {
            // each worker's OpenMP regions get their share of the host cores
            omp_set_num_threads(std::max(1, hostThreads / nbWorkers));
            try
            {
                AVDM_LOG_INFO("Worker " << w << " of " << nbWorkers << " uses device " << devices[w] << ": " << share[w].size() << " cameras.");
                if(!allViews.empty())
                    gpujob.computeShared(w, devices[w], share[w], allViews, exchange);
                else if(!share[w].empty())
                    gpujob.compute(devices[w], share[w]);
            }
            catch(...)
            {
                exchange.fail(std::current_exception()); // wake the workers waiting for a view of this one
                std::lock_guard<std::mutex> lock(errorGuard);
                if(!firstError)
                    firstError = std::current_exception();
            }
        });
    for(std::thread& t : threads)
        t.join();
    if(!allViews.empty())
        AVDM_LOG_INFO("Pyramid exchange: " << exchange.nbBuilt << " views converted once, " << exchange.nbCopied << " peer copies ("
                                           << (double)exchange.bytesCopied / (1024.0 * 1024.0) << " MB) instead of decoding them again; "
                                           << exchange.nbDeclined << " views over the residency budget were decoded where needed.");
    if(firstError)
        std::rethrow_exception(firstError);
}

} // namespace avdm_host
