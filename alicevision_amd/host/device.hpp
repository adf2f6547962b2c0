// device.hpp — HIP-side plumbing of the host layer: error checks, RAII device / pinned-host buffers, streams, and the
// per-device LRU cache of Lab pyramids and camera-parameter blocks.
// Restates the roles of depthMap/cuda/host/{memory.hpp:326-548 (CudaDeviceMemoryPitched / CudaHostMemoryHeap), utils.hpp:10-40
// (CHECK_CUDA_ERROR), DeviceStreamManager.{hpp,cpp}, LRUCache.hpp:30-136, DeviceCache.{hpp,cpp}} for HIP and the avdm C ABI:
// buffers are plain hipMalloc allocations with explicit pitches (the kernels take (pointer, pitch)), camera parameters are
// host-side avdm_camera_t blocks passed by pointer (no constant-memory slots, hence no 100-slot limit), images are
// avdm_pyramid_t.
#pragma once

#include "MultiViewParams.hpp"

#include <avdm.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <exception>
#include <list>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace avdm_host {

#define AVDM_HIP_CHECK(expr)                                                                                                                                  \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        const hipError_t avdm_hip_err = (expr);                                                                                                               \
        if(avdm_hip_err != hipSuccess)                                                                                                                        \
            throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(avdm_hip_err) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + \
                                     ")");                                                                                                                    \
    } while(0)

// every avdm_* entry point returns 0 or an error code with a message (include/avdm.h)
inline void avdmCheck(int status, const char* what)
{
    if(status != 0)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(status) + "): " + avdm_last_error());
}

struct JpegImage;
// a JPEG's coefficients -> linear float RGBA (pitch width * 16) on the device: avdm_image_decode_jpeg, then avdm_image_decode_integer with the
// sRGB decoding (image::readImage(path, img, LINEAR) for an 8-bit file).  Synchronises `stream` (its temporaries die on return).
void decodeJpegToLinearRgba(const JpegImage& jpeg, float* rgba_d, hipStream_t stream);

// One hipMalloc for the many fixed-size buffers of a worker (no counterpart in the reference, which allocates every CudaDeviceMemoryPitched
// by itself, memory.hpp:326-548).  A worker's set-up creates ~12 device buffers per tile slot — ~500 hipMalloc calls for the default tiling of a
// 12 MP camera, and as many hipFree calls (each a device-wide wait) at the end: measured 0.3 s + 0.4 s of a 6.4 s job.  While
// a DeviceArena::Scope is alive on a thread, DeviceBuffer::allocate on THAT thread takes its bytes from the arena (4 KiB-aligned, one guard
// page between neighbours); a request the arena cannot serve falls back to hipMalloc.  The arena must outlive the buffers carved from it.
class DeviceArena
{
  public:
    DeviceArena() = default;
    explicit DeviceArena(size_t bytes)
    {
        if(bytes && hipMalloc(&_base, bytes) == hipSuccess)
            _bytes = bytes;
        else
            (void)hipGetLastError(); // no arena: every buffer allocates for itself
    }
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    ~DeviceArena() { release(); }
    // (every buffer carved from it must be gone)
    void release()
    {
        if(_base)
            (void)hipFree(_base);
        _base = nullptr, _bytes = _used = 0;
    }
    void* take(size_t bytes)
    {
        const size_t need = ((bytes + 4095) & ~(size_t)4095) + 4096;
        if(_base == nullptr || _used + need > _bytes)
            return nullptr;
        void* p = static_cast<char*>(_base) + _used;
        _used += need;
        return p;
    }
    size_t bytes() const { return _bytes; }
    size_t used() const { return _used; }
    static DeviceArena*& current()
    {
        static thread_local DeviceArena* a = nullptr;
        return a;
    }
    struct Scope
    {
        DeviceArena* prev;
        explicit Scope(DeviceArena* a) : prev(current()) { current() = a; }
        ~Scope() { current() = prev; }
        Scope(const Scope&) = delete;
        Scope& operator=(const Scope&) = delete;
    };

  private:
    void* _base = nullptr;
    size_t _bytes = 0, _used = 0;
};

class DeviceBuffer
{
  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) { allocate(bytes); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : _p(o._p), _bytes(o._bytes), _owned(o._owned) { o._p = nullptr, o._bytes = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept
    {
        if(this != &o)
        {
            release();
            _p = o._p, _bytes = o._bytes, _owned = o._owned;
            o._p = nullptr, o._bytes = 0;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void allocate(size_t bytes)
    {
        release();
        if(bytes)
        {
            DeviceArena* const arena = DeviceArena::current();
            _p = arena != nullptr ? arena->take(bytes) : nullptr;
            _owned = _p == nullptr;
            if(_owned)
                AVDM_HIP_CHECK(hipMalloc(&_p, bytes));
        }
        _bytes = bytes;
    }
    void release()
    {
        if(_p && _owned)
            (void)hipFree(_p);
        _p = nullptr, _bytes = 0, _owned = true;
    }
    void* ptr() const { return _p; }
    template <typename T>
    T* as() const { return static_cast<T*>(_p); }
    size_t bytes() const { return _bytes; }

  private:
    void* _p = nullptr;
    size_t _bytes = 0;
    bool _owned = true; // false: carved from a DeviceArena
};

// pinned, unpadded host buffer (CudaHostMemoryHeap)
template <typename T>
class PinnedBuffer
{
  public:
    PinnedBuffer() = default;
    explicit PinnedBuffer(size_t n) { allocate(n); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    PinnedBuffer(PinnedBuffer&& o) noexcept : _p(o._p), _n(o._n) { o._p = nullptr, o._n = 0; }
    ~PinnedBuffer()
    {
        if(_p)
            (void)hipHostFree(_p);
    }
    void allocate(size_t n)
    {
        if(_p)
            (void)hipHostFree(_p);
        _p = nullptr;
        if(n)
            AVDM_HIP_CHECK(hipHostMalloc((void**)&_p, n * sizeof(T), hipHostMallocDefault));
        _n = n;
    }
    T* data() const { return _p; }
    size_t size() const { return _n; }
    T& operator[](size_t i) const { return _p[i]; }

  private:
    T* _p = nullptr;
    size_t _n = 0;
};

// DeviceStreamManager.cpp: n non-blocking streams, round-robin access
class DeviceStreamManager
{
  public:
    explicit DeviceStreamManager(int nbStreams);
    ~DeviceStreamManager();
    // give the streams (and the library's per-stream scratch blocks) back before the object dies; getStream() must not be called afterwards
    void destroy();
    hipStream_t getStream(int i) const { return _streams.at(i % _streams.size()); }
    int getNbStreams() const { return (int)_streams.size(); }
    void waitStream(int i) const { AVDM_HIP_CHECK(hipStreamSynchronize(getStream(i))); }

  private:
    std::vector<hipStream_t> _streams;
};

// DeviceMipmapImage.{hpp,cpp}: the fp16 Lab pyramid of one camera in HBM
class DeviceMipmapImage
{
  public:
    // DeviceMipmapImage.cpp:28-90 via avdm_pyramid_layout + avdm_pyramid_fill (x255 -> fp16 -> Gaussian downscale -> Lab -> levels)
    // staging: a device buffer kept by the caller for the uploaded float RGBA image (grown when too small) — without it every call allocates
    // and frees one (a hipFree is a device-wide wait: with several views in flight on several streams each would wait for all the others)
    void fill(const HostImage& img, int minDownscale, int maxDownscale, int filterMode, hipStream_t stream, DeviceBuffer* staging = nullptr);
    // a copy of a pyramid that lives on another device (or on this one): hipMemcpyPeerAsync over xGMI, same bytes, own descriptor
    void copyFromPeer(const DeviceMipmapImage& src, int srcDevice, int dstDevice, hipStream_t stream);
    const avdm_pyramid_t& pyramid() const { return _pyr; }
    size_t bytes() const { return _buf.bytes(); }

  private:
    avdm_pyramid_t _pyr{};
    DeviceBuffer _buf;
};

// LRUCache.hpp:30-136: fixed number of slots; insert() returns the slot and whether the key is new (evicting the least
// recently used key when full)
template <typename Key>
class LRUCache
{
  public:
    explicit LRUCache(int maxSize) : _max(maxSize) {}
    bool insert(const Key& key, int* slot)
    {
        auto it = _map.find(key);
        if(it != _map.end())
        {
            _order.splice(_order.begin(), _order, it->second.second);
            *slot = it->second.first;
            return false;
        }
        int s;
        if((int)_map.size() < _max)
            s = (int)_map.size();
        else
        {
            const Key victim = _order.back();
            s = _map[victim].first;
            _map.erase(victim);
            _order.pop_back();
        }
        _order.push_front(key);
        _map[key] = {s, _order.begin()};
        *slot = s;
        return true;
    }
    bool find(const Key& key, int* slot) const
    {
        auto it = _map.find(key);
        if(it == _map.end())
            return false;
        *slot = it->second.first;
        return true;
    }

  private:
    int _max;
    std::list<Key> _order;
    std::map<Key, std::pair<int, typename std::list<Key>::iterator>> _map;
};

// PyramidExchange — the multi-GPU design of this stage (BASELINE north_star; no counterpart in the reference, where every device
// thread decodes, uploads and converts every image it needs by itself, DepthMapEstimator.cpp:224-232 + DeviceCache.cpp:222-281).
// The workers of one process (one host thread per GPU) share it: the views the job needs are partitioned over the workers
// (view index modulo the number of workers); the OWNER of a view decodes it, builds its Lab pyramid once and keeps it resident;
// every other worker that needs the view copies the finished pyramid device-to-device (hipMemcpyPeerAsync: xGMI between the GPUs of
// a node) into its own LRU cache.  A 12 MP pyramid is 128 MB = ~1 ms over one xGMI link against ~50 ms of EXR decode + 3 ms of PCIe
// upload + conversion, and each image is decoded once per node instead of once per device that uses it.
class PyramidExchange
{
  public:
    // devices[w] = physical device of worker w (the same device may appear several times: that is how the one-GPU tests run two workers)
    explicit PyramidExchange(std::vector<int> devices) : _devices(std::move(devices)), _residentBytes(_devices.size(), 0) {}
    int nbWorkers() const { return (int)_devices.size(); }
    int deviceOf(int worker) const { return _devices.at(worker); }
    int ownerOf(int camId) const { return camId % nbWorkers(); }
    // Residency is BOUNDED: an owner keeps at most `budget` bytes of published pyramids on its device (default: a quarter of the device's
    // memory, AVDM_EXCHANGE_BUDGET_MB overrides).  A view that no longer fits is DECLINED: every worker that needs it — its owner
    // included — decodes and converts it itself into its own LRU cache, which is what the reference does for every neighbour on
    // every device (DepthMapEstimator.cpp:224-232, nbRcPerBatch * (1 + maxTCams) pyramids at a time).
    void setBudgetBytes(size_t perWorker) { _budget = perWorker; }
    size_t budgetBytes() const { return _budget; }
    size_t residentBytes(int worker);
    // owner side: hand over a finished pyramid; it stays resident until the exchange dies.  false = over budget: the view is declined
    bool publish(int camId, std::shared_ptr<const DeviceMipmapImage> img);
    // owner side: this view will not be published (over budget, or the owner's pre-pass is over and it never came up)
    void decline(int camId);
    bool isDeclined(int camId);
    // the owner's resident pyramid, waiting until it is published or declined (nullptr); rethrows the failure of a worker that died first
    std::shared_ptr<const DeviceMipmapImage> await(int camId);
    std::shared_ptr<const DeviceMipmapImage> find(int camId);
    void fail(std::exception_ptr e);
    std::atomic<long> nbDeclined{0};
    // statistics for the log: pyramids built by their owners / copied between workers, bytes copied
    std::atomic<long> nbBuilt{0}, nbCopied{0};
    std::atomic<long long> bytesCopied{0};

  private:
    std::vector<int> _devices;
    std::mutex _mutex;
    std::condition_variable _published;
    std::map<int, std::shared_ptr<const DeviceMipmapImage>> _resident;
    std::set<int> _declined;
    std::vector<size_t> _residentBytes; // per owner
    size_t _budget = (size_t)-1;
    std::exception_ptr _failure;
};

// DeviceCache.{hpp,cpp}: one instance per device (owned by the DepthMapEstimator::compute call of that device)
class DeviceCache
{
  public:
    DeviceCache(int maxMipmapImages, int maxCameraParams, int filterMode);
    // multi-GPU: views owned by other workers are copied from their owners instead of decoded again (see PyramidExchange)
    void setExchange(PyramidExchange* exchange, int worker) { _exchange = exchange, _worker = worker; }
    // owner side of the exchange: decode + build + publish a view this worker owns (no LRU slot is taken: the pyramid stays resident)
    void buildOwnedView(int camId, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp, hipStream_t stream);
    // DeviceCache.cpp:222-281
    void addMipmapImage(int camId, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp, hipStream_t stream);
    // the same for a list of views at once: LRU slots taken in list order on the calling thread, then one host thread per NEW view decodes it,
    // uploads it and converts it on a stream of its own (pyramid buffers of evicted slots and the upload staging buffers are reused).  Without
    // the exchange only (the multi-GPU path publishes its views in the pre-pass).
    void addMipmapImages(const std::vector<int>& camIds, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp);
    // DeviceCache.cpp:283-323 (+ fillHostCameraParameters :41-134 through avdm_camera_fill)
    void addCameraParams(int camId, int downscale, const MultiViewParams& mp);
    // DeviceCache.cpp:325-365: throw when absent
    const DeviceMipmapImage& requestMipmapImage(int camId, const MultiViewParams& mp) const;
    const avdm_camera_t& requestCameraParams(int camId, int downscale, const MultiViewParams& mp) const;

    // where this worker's image time went (seconds on its host thread): one log line per worker makes a multi-GPU run diagnosable
    struct ImageTimes
    {
        double awaitOwner = 0.0;   // blocked until another worker had published the view
        double peerCopy = 0.0;     // hipMemcpyPeerAsync + wait
        double localBuild = 0.0;   // decode (unless cached) + upload + pyramid on this device
        int received = 0, built = 0;
        long long bytesReceived = 0;
    };
    const ImageTimes& imageTimes() const { return _times; }
    // free the pyramids of the slots and the upload buffers now (the cache stays usable: empty)
    void releaseImages();

  private:
    ImageTimes _times;
    int _filterMode;
    LRUCache<int> _mipmapCache;
    LRUCache<std::pair<int, int>> _cameraParamCache;
    std::vector<std::shared_ptr<const DeviceMipmapImage>> _mipmaps; // a slot owns its pyramid, or aliases one this worker published
    std::vector<avdm_camera_t> _cameraParams;
    std::vector<std::unique_ptr<DeviceBuffer>> _staging; // upload buffers of addMipmapImages, one per team thread
    PyramidExchange* _exchange = nullptr;
    int _worker = 0;
};

void logDeviceMemoryInfo();
void getDeviceMemoryInfo(double& availableMB, double& usedMB, double& totalMB);

} // namespace avdm_host
