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
                                           << (double)exchange.bytesCopied / (1024.0 * 1024.0) << " MB) instead of decoding them again.");
    if(firstError)
        std::rethrow_exception(firstError);
}

} // namespace avdm_host
