// SgmDepthList.cpp — depth-plane list of one R-camera tile (behaviour of depthMap/SgmDepthList.cpp of the
// reference, file:line cited per step; written from scratch around a per-camera SEED INDEX).
//
// What the stage computes: the distances (along R's optical axis) of the fronto-parallel planes the sweep visits, and per T camera
// the sub-range of those planes worth sweeping.  Inputs are the SfM landmarks seen by R ("seeds") and, per T camera, the depths
// obtained by walking T's epipolar line of R's reference pixel.
//
// Design (not the reference's): the reference rescans every landmark of the scene once per tile for the range statistics and once
// more per (tile, T camera) for the common-observation range (SgmDepthList.cpp:277-415) — with the default 20 tiles x 10 T cameras
// of a 12 MP view that is 220 passes over the landmark map, serialised between asynchronous launches.  Here each R camera gets ONE
// pass: `SeedIndex` keeps, in landmark order, the plane distance (float and double, as the two consumers need), the R observation
// and a bit set of the observing cameras; tiles and T cameras query it.  An LRU of a few cameras keeps it alive across the tiles of
// a camera and across the batches of DepthMapEstimator.  Arithmetic (float / double roles, accumulation order) is the reference's,
// so the plans are identical (tests/test_host_cpu.py against oracle/host_oracle.py).
#include "SgmDepthList.hpp"

#include "log.hpp"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <list>
#include <memory>
#include <mutex>
#include <sstream>

namespace avdm_host {

int indexOfNearestSorted(const std::vector<float>& in_vector, const float value)
{
    // first element >= value, then the closer of it and its predecessor (ties go to the upper one); -1 past the end (SgmDepthList.cpp:25-42)
    const auto ge = std::lower_bound(in_vector.begin(), in_vector.end(), value);
    if(ge == in_vector.end())
        return -1;
    if(ge != in_vector.begin() && (value - *(ge - 1)) < (*ge - value))
        return (int)(ge - in_vector.begin()) - 1;
    return (int)(ge - in_vector.begin());
}

namespace {

// ---- fronto-parallel frame of a camera: plane through the centre, normal = optical axis ----
struct AxisFrame
{
    Point3d origin, axis;
    AxisFrame(const MultiViewParams& mp, int cam) : origin(mp.CArr[cam]), axis((mp.iRArr[cam] * Point3d(0.0, 0.0, 1.0)).normalize()) {}
    double absDistance(const Point3d& X) const { return pointPlaneDistance(X, origin, axis); }
    double signedDistance(const Point3d& X) const { return orientedPointPlaneDistance(X, origin, axis); }
    Point3d at(float depth) const { return origin + axis * depth; }
};

// ---- boost::accumulators::tail_quantile (tail cache of N samples; not in the reference tree, published behaviour): the cache keeps the N
//      most extreme samples ordered from the extreme inwards; quantile(p) = cache[ceil(n * q) - 1] when that index is cached, else NaN ----
class ExtremeTail
{
  public:
    ExtremeTail(bool lowTail, std::size_t capacity) : _low(lowTail), _cap(capacity) {}
    void add(float v)
    {
        ++_seen;
        const auto pos = _low ? std::upper_bound(_kept.begin(), _kept.end(), v) : std::upper_bound(_kept.begin(), _kept.end(), v, std::greater<float>());
        if(_kept.size() < _cap)
            _kept.insert(pos, v);
        else if(pos != _kept.end())
        {
            _kept.insert(pos, v);
            _kept.pop_back();
        }
    }
    float quantile(double p) const
    {
        const std::size_t rank = (std::size_t)std::ceil(_seen * (_low ? p : 1.0 - p));
        return (rank >= 1 && rank < _kept.size()) ? _kept[rank - 1] : std::numeric_limits<float>::quiet_NaN();
    }

  private:
    bool _low;
    std::size_t _cap, _seen = 0;
    std::vector<float> _kept;
};

// ---- the seeds of one R camera: every landmark R observes, in landmark-id order ----
struct Seed
{
    Point3d X;
    double distance;        // |distance| to R's fronto-parallel plane through its centre (double: the per-T range)
    float distanceF;        // the same, rounded to float first like the range statistics do (SgmDepthList.cpp:306)
    unsigned obsX, obsY;    // R's observation in full-size pixels, truncated like `ROI::contains(unsigned, unsigned)` receives it
    std::size_t seenBy;     // offset of this seed's camera bit set in SeedIndex::bits
};
struct SeedIndex
{
    const MultiViewParams* mp;
    unsigned long long mpGeneration; // a new MultiViewParams allocated at the address of a dead one is not the same scene
    int rc;
    std::vector<Seed> seeds;
    std::vector<std::uint64_t> bits; // per seed: ceil(nbCameras / 64) words, bit c = camera index c observes the landmark
    std::size_t wordsPerSeed = 0;
    bool seenByCam(const Seed& s, int cam) const { return (bits[s.seenBy + (std::size_t)cam / 64] >> ((unsigned)cam % 64)) & 1ull; }
};

std::shared_ptr<const SeedIndex> buildSeedIndex(const MultiViewParams& mp, int rc)
{
    auto idx = std::make_shared<SeedIndex>();
    idx->mp = &mp;
    idx->mpGeneration = mp.generation();
    idx->rc = rc;
    const int nCams = mp.getNbCameras();
    idx->wordsPerSeed = ((std::size_t)nCams + 63) / 64;
    std::map<IndexT, int> camOfView;
    for(int c = 0; c < nCams; ++c)
        camOfView[mp.getViewId(c)] = c;
    const IndexT rcView = mp.getViewId(rc);
    const AxisFrame frame(mp, rc);
    for(const auto& kv : mp.getInputSfMData().landmarks)
    {
        const Landmark& lm = kv.second;
        const auto mine = lm.observations.find(rcView);
        if(mine == lm.observations.end())
            continue;
        Seed s;
        s.X = lm.X;
        s.distance = frame.absDistance(lm.X);
        s.distanceF = static_cast<float>(s.distance);
        s.obsX = (unsigned int)mine->second.x;
        s.obsY = (unsigned int)mine->second.y;
        s.seenBy = idx->bits.size();
        idx->bits.resize(idx->bits.size() + idx->wordsPerSeed, 0);
        for(const auto& ob : lm.observations)
        {
            const auto cam = camOfView.find(ob.first);
            if(cam != camOfView.end())
                idx->bits[s.seenBy + (std::size_t)cam->second / 64] |= 1ull << ((unsigned)cam->second % 64);
        }
        idx->seeds.push_back(s);
    }
    return idx;
}

// a handful of cameras stay indexed: the tiles of a camera are consecutive, batches hold a few R cameras
std::shared_ptr<const SeedIndex> seedIndexOf(const MultiViewParams& mp, int rc)
{
    static std::mutex guard;
    static std::list<std::shared_ptr<const SeedIndex>> recent;
    std::lock_guard<std::mutex> lock(guard);
    for(auto it = recent.begin(); it != recent.end(); ++it)
        if((*it)->mp == &mp && (*it)->mpGeneration == mp.generation() && (*it)->rc == rc)
        {
            recent.splice(recent.begin(), recent, it);
            return recent.front();
        }
    recent.push_front(buildSeedIndex(mp, rc));
    if(recent.size() > 8)
        recent.pop_back();
    return recent.front();
}

// range statistics of the seeds a tile uses (SgmDepthList.cpp:277-345): 0.1 % / 99.9 % tail quantiles, distance of the mean point, count
struct SeedStats
{
    std::size_t count = 0;
    float nearQ = 0.f, farQ = 0.f, mid = 0.f;
};

} // namespace

// =====================================================================================================================
struct SgmDepthListScratch // per computeListRc() call
{
    std::shared_ptr<const SeedIndex> index;
    ROI fullRoi;
    bool perTile;
    bool uses(const Seed& s) const { return !perTile || fullRoi.contains(s.obsX, s.obsY); }
};

static SeedStats seedStatistics(const SgmDepthListScratch& S, const AxisFrame& frame, double percentile)
{
    SeedStats st;
    ExtremeTail nearTail(true, 1000), farTail(false, 1000);
    Point3d sum; // mean of the landmark positions, accumulated in landmark order
    for(const Seed& s : S.index->seeds)
    {
        if(!S.uses(s))
            continue;
        nearTail.add(s.distanceF);
        farTail.add(s.distanceF);
        sum = sum + s.X;
        ++st.count;
    }
    if(st.count > 0)
    {
        st.nearQ = nearTail.quantile(1.0 - percentile);
        st.farQ = farTail.quantile(percentile);
        st.mid = (float)frame.absDistance(sum / static_cast<float>(st.count));
    }
    return st;
}

// depth interval covered by the landmarks R and `tc` both observe (SgmDepthList.cpp:347-415); false when they share none
static bool commonSeedRange(const SgmDepthListScratch& S, int tc, double& zNear, double& zFar)
{
    zNear = std::numeric_limits<double>::max();
    zFar = std::numeric_limits<double>::min();
    for(const Seed& s : S.index->seeds)
        if(S.index->seenByCam(s, tc) && S.uses(s))
        {
            zNear = std::min(zNear, s.distance);
            zFar = std::max(zFar, s.distance);
        }
    return zNear <= zFar;
}

// =====================================================================================================================
void SgmDepthList::computeListRc()
{
    const auto t0 = std::chrono::steady_clock::now();
    _depths.clear();
    _depthsTcLimits.clear();
    AVDM_LOG_DEBUG(_tile << "Compute SGM depths list.");

    float nearObs, farObs, midObs;
    std::size_t nbObs;
    getMinMaxMidNbDepthFromSfM(nearObs, farObs, midObs, nbObs);
    if(nbObs < 2) // :61-65 — the tile is skipped (written invalid) by the caller
    {
        AVDM_LOG_INFO(_tile << "Cannot get min/max/middle depth from SfM.");
        return;
    }

    // candidate depths per T camera: the epipolar walk, or the pixel-size ladder when it yields too little (:67-85)
    const std::size_t nT = _tile.sgmTCams.size();
    std::vector<std::vector<float>> perT(nT);
    for(std::size_t c = 0; c < nT; ++c)
    {
        computeRcTcDepths(_tile.sgmTCams[c], nbObs < 10 ? -1.f : midObs, perT[c]);
        if(perT[c].size() < 10)
        {
            AVDM_LOG_DEBUG(_tile << "Not enough valid samples over the epipolar line. Compute depth list from R camera pixel size.");
            perT[c].clear();
            computePixelSizeDepths(nearObs, midObs, farObs * (float)_sgmParams.prematchingMaxDepthScale, perT[c]);
        }
    }

    // overall interval of the candidates (:87-104; note numeric_limits<float>::min() is the smallest POSITIVE float, kept)
    float lo = std::numeric_limits<float>::max(), hi = std::numeric_limits<float>::min();
    for(const auto& list : perT)
        for(const float d : list)
        {
            lo = std::min(lo, d);
            hi = std::max(hi, d);
        }
    if(lo > hi)
    {
        AVDM_LOG_INFO(_tile << "No depths found.");
        return;
    }

    // narrowed to the seeds' interval inflated by seedsRangeInflate when the two overlap (:106-137)
    float first = lo, last = hi;
    if(_sgmParams.useSfmSeeds && !_mp.getInputSfMData().landmarks.empty() && nbObs > 10)
    {
        const float margin = _sgmParams.seedsRangeInflate * (farObs - nearObs);
        first = std::max(0.f, nearObs - margin);
        last = farObs + margin;
        const bool disjoint = hi < first || lo > last;
        if(!disjoint)
        {
            first = std::max(lo, first);
            last = std::min(hi, last);
        }
        AVDM_LOG_DEBUG(_tile << "Final depth range (intersection: frustums / landmarks with margin): [" << first << "-" << last << "]");
    }

    // the list itself, thinned by re-stepping when it exceeds maxDepths (:139-156)
    computeRcDepthList(first, last, _sgmParams.stepZ > 0.0f ? (float)_sgmParams.stepZ : 1.0f, perT);
    if(_sgmParams.maxDepths > 0 && (int)_depths.size() > _sgmParams.maxDepths)
    {
        const float stretch = float(_depths.size()) / float(_sgmParams.maxDepths);
        AVDM_LOG_DEBUG(_tile << "Too many depths (" << _depths.size() << " > " << _sgmParams.maxDepths << "): re-stepping with factor " << stretch);
        computeRcDepthList(first, last, stretch, perT);
        if((int)_depths.size() > _sgmParams.maxDepths)
            _depths.resize(_sgmParams.maxDepths);
    }

    // per T camera: [index of the plane nearest to its first candidate, number of planes up to its last one] (:163-187)
    _depthsTcLimits.assign(nT, Pixel(-1, -1));
    for(std::size_t c = 0; c < nT && !perT.empty(); ++c)
    {
        int a = indexOfNearestSorted(_depths, perT[c].front());
        int b = indexOfNearestSorted(_depths, perT[c].back());
        a = a < 0 ? 0 : a;
        b = b < 0 ? (int)_depths.size() - 1 : b;
        _depthsTcLimits[c] = Pixel(a, b - a + 1);
    }
    if(_sgmParams.exportDepthsTxtFiles)
        exportTxtFiles(perT);

    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    AVDM_LOG_DEBUG(_tile << "Depth list: " << _depths.size() << " planes in [" << first << "-" << last << "] from " << nbObs << " seeds, " << ms
                         << " ms.");
}

void SgmDepthList::removeTcWithNoDepth(Tile& tile)
{
    // compacts the T cameras of the tile (and their limits) to those with a plane range (:194-221)
    assert(tile.rc == _tile.rc);
    std::size_t kept = 0;
    for(std::size_t c = 0; c < tile.sgmTCams.size(); ++c)
    {
        const Pixel lim = _depthsTcLimits.at(c);
        if(lim.x == -1 || lim.y == -1)
        {
            AVDM_LOG_INFO(_tile << "Remove T camera (tc: " << tile.sgmTCams[c] << ", view id: " << _mp.getViewId(tile.sgmTCams[c]) << ") no depth found.");
            continue;
        }
        tile.sgmTCams[kept] = tile.sgmTCams[c];
        _depthsTcLimits[kept] = lim;
        ++kept;
    }
    tile.sgmTCams.resize(kept);
    _depthsTcLimits.resize(kept);
}

void SgmDepthList::logRcTcDepthInformation() const
{
    std::ostringstream o;
    o << "Camera / Depth information: " << std::endl
      << "\t- R camera:" << std::endl
      << "\t   - id: " << _tile.rc << std::endl
      << "\t   - view id: " << _mp.getViewId(_tile.rc) << std::endl
      << "\t   - depth planes: " << _depths.size() << std::endl
      << "\t   - depths range: [" << _depths.front() << "-" << _depths.back() << "]" << std::endl
      << "\t- T cameras:" << std::endl;
    for(std::size_t c = 0; c < _tile.sgmTCams.size(); ++c)
    {
        const Pixel lim = _depthsTcLimits[c];
        o << "\t   - T camera (" << (c + 1) << "/" << _tile.sgmTCams.size() << "):" << std::endl
          << "\t      - id: " << _tile.sgmTCams[c] << std::endl
          << "\t      - view id: " << _mp.getViewId(_tile.sgmTCams[c]) << std::endl
          << "\t      - depth planes: " << lim.y << std::endl
          << "\t      - depths range: [" << _depths[lim.x] << "-" << _depths[lim.x + lim.y - 1] << "]" << std::endl
          << "\t      - depth indexes range: [" << lim.x << "-" << lim.x + lim.y << "]" << std::endl;
    }
    AVDM_LOG_INFO(_tile << o.str());
}

void SgmDepthList::checkStartingAndStoppingDepth() const
{
    // assertions only in the reference (:250-275): the union of the T ranges starts at plane 0 and stays inside the list
    int from = std::numeric_limits<int>::max(), to = 0;
    for(const Pixel& lim : _depthsTcLimits)
    {
        from = std::min(from, lim.x);
        to = std::max(to, lim.x + lim.y);
    }
    if(!_depthsTcLimits.empty() && (from != 0 || to > (int)_depths.size()))
        AVDM_LOG_DEBUG(_tile << "Depth limits: starting depth index " << from << ", stopping depth index " << to << " / " << _depths.size());
}

void SgmDepthList::getMinMaxMidNbDepthFromSfM(float& out_min, float& out_max, float& out_mid, std::size_t& out_nbDepths) const
{
    SgmDepthListScratch S{seedIndexOf(_mp, _tile.rc), upscaleROI(_tile.roi, (float)_mp.getProcessDownscale()), _sgmParams.depthListPerTile};
    const SeedStats st = seedStatistics(S, AxisFrame(_mp, _tile.rc), _sgmParams.seedsRangePercentile);
    out_min = st.nearQ;
    out_max = st.farQ;
    out_mid = st.mid;
    out_nbDepths = st.count;
    AVDM_LOG_DEBUG(_tile << "Seeds of the R camera (view id " << _mp.getViewId(_tile.rc) << "): " << st.count << " landmarks, depth quantiles ["
                         << st.nearQ << "-" << st.farQ << "] at " << _sgmParams.seedsRangePercentile << ", mid " << st.mid);
}

void SgmDepthList::getRcTcDepthRangeFromSfM(int tc, double& out_zmin, double& out_zmax) const
{
    SgmDepthListScratch S{seedIndexOf(_mp, _tile.rc), upscaleROI(_tile.roi, (float)_mp.getProcessDownscale()), _sgmParams.depthListPerTile};
    if(!commonSeedRange(S, tc, out_zmin, out_zmax)) // :399-403
        AVDM_THROW_ERROR(_tile << "Cannot compute min/max depth from common Rc/Tc SfM observations." << std::endl
                               << "No common observations found (tc view id: " << _mp.getViewId(tc) << ").");
    AVDM_LOG_DEBUG(_tile << "Common seeds of rc " << _tile.rc << " / tc " << tc << ": depth range [" << out_zmin << "-" << out_zmax << "]");
}

// Walk T's epipolar line of R's reference pixel at SGM-scale pixel steps, triangulate each position with the reference ray and keep the
// strictly increasing positive depths whose ray angle is admissible (:417-545).  midDepth < 0: fewer than 10 seeds, the direction probe
// then uses the point "behind" the camera exactly like the reference.
void SgmDepthList::computeRcTcDepths(int tc, float midDepth, std::vector<float>& out_depths) const
{
    const int rc = _tile.rc;
    const AxisFrame frame(_mp, rc);
    const Point2d refPix = _sgmParams.depthListPerTile
                             ? Point2d(_tile.roi.x.begin + (_tile.roi.width() * 0.5), _tile.roi.y.begin + (_tile.roi.height() * 0.5))
                             : Point2d(_mp.getWidth(rc) * 0.5, _mp.getHeight(rc) * 0.5);

    // the segment of the epipolar line to visit: between the images of the reference ray at the nearest / farthest common seed,
    // clipped to T's image (both ends stay at the origin when the line misses the image)
    Point2d midInT, segA, segB;
    {
        Point3d C;
        Matrix3x3 R, iR, K, iK, iP;
        _mp.decomposeProjectionMatrix(C, R, iR, K, iK, iP, _mp.camArr[rc]);
        const Point3d ray = iP * refPix;
        _mp.getPixelFor3DPoint(&midInT, (ray * midDepth) + C, _mp.camArr[tc]);
        double zNear, zFar;
        getRcTcDepthRangeFromSfM(tc, zNear, zFar);
        Point2d nearInT, farInT;
        _mp.getPixelFor3DPoint(&nearInT, (ray * zNear) + C, _mp.camArr[tc]);
        _mp.getPixelFor3DPoint(&farInT, (ray * zFar) + C, _mp.camArr[tc]);
        get2dLineImageIntersection(&segA, &segB, nearInT, farInT, _mp, tc);
    }
    const int nPix = static_cast<int>((segB - segA).size());
    const int nSteps = nPix / _sgmParams.scale;
    const Point2d stride = (segB - segA).normalize() * std::max(1.0, double(_sgmParams.scale));

    // which way along the segment do depths grow?  probe one stride from the mid point
    bool forward = true;
    {
        Point3d X;
        if(!triangulateMatch(X, refPix, midInT, rc, tc, _mp))
            return;
        const float d0 = (float)frame.signedDistance(X);
        if(!triangulateMatch(X, refPix, midInT + stride, rc, tc, _mp))
            return;
        forward = !(d0 > (float)frame.signedDistance(X));
    }

    out_depths.reserve(std::max(nSteps, 0));
    const Point3d refRay = _mp.iCamArr[rc] * refPix;
    const Point2d start = forward ? segA : segB;
    const double sense = forward ? 1.0 : -1.0;
    float floorDepth = -1.0f; // the next depth must exceed this
    for(int i = 0; i < nSteps; ++i)
    {
        const Point2d q = start + (stride * double(i) * sense);
        if(!_mp.isPixelInImage(q, tc))
            continue;
        const float angle = (float)angleBetwV1andV2(refRay, _mp.iCamArr[tc] * q);
        if(angle < _mp.getMinViewAngle() || angle > _mp.getMaxViewAngle())
            continue;
        Point3d X;
        if(!triangulateMatch(X, refPix, q, rc, tc, _mp))
            continue;
        const float d = float(frame.signedDistance(X));
        if(d > 0.0f && d > floorDepth)
        {
            out_depths.push_back(d);
            floorDepth = d + std::numeric_limits<float>::epsilon();
        }
    }
    out_depths.shrink_to_fit();
    AVDM_LOG_DEBUG(_tile << "Epipolar walk rc " << rc << " / tc " << tc << ": " << nPix << " px, " << nSteps << " steps at SGM scale, " << out_depths.size()
                         << " depths" << (out_depths.empty() ? std::string() : " in [" + std::to_string(out_depths.front()) + "-" + std::to_string(out_depths.back()) + "]"));
}

// Ladder of depths whose rungs are one R pixel (at 6 x SGM scale) apart in depth, grown from the mid depth outwards until the observed
// range is covered, at most 2048 rungs (:547-625)
void SgmDepthList::computePixelSizeDepths(float minObsDepth, float midObsDepth, float maxObsDepth, std::vector<float>& out_depths) const
{
    const int kHalfBudget = 1024;
    const float pixelSpan = float(_sgmParams.scale) * 6.0f;
    const AxisFrame frame(_mp, _tile.rc);
    auto rung = [&](float depth) { return (float)_mp.getCamPixelSize(frame.at(depth), _tile.rc, pixelSpan); };

    int up = 0;
    float top = midObsDepth;
    for(; top < maxObsDepth && up < kHalfBudget; ++up)
        top += rung(top);
    int down = 0;
    float bottom = midObsDepth;
    for(; bottom > minObsDepth && down < 2 * kHalfBudget - up; ++down)
        bottom -= rung(bottom);

    float step = 1.0f;
    int n = 0;
    for(float d = bottom; d < top && step > 0.0f && n < 2 * kHalfBudget; ++n)
    {
        out_depths.push_back(d);
        step = rung(d);
        d += step;
    }
    for(std::size_t i = 1; i < out_depths.size(); ++i)
        if(!(out_depths[i - 1] < out_depths[i]))
            throw std::runtime_error("getDepthsByPixelSize not asc.");
}

// From `firstDepth`, advance by the finest local spacing any T camera's candidates have around the current depth, times `scaleFactor`,
// until `lastDepth` (:627-660).  A candidate list contributes only where it has a successor (the reference compares an int index with
// size() - 1 as unsigned: the "-1 = past the end" result of the search is skipped by that same test).
void SgmDepthList::computeRcDepthList(float firstDepth, float lastDepth, float scaleFactor, const std::vector<std::vector<float>>& dephtsPerTc)
{
    _depths.clear();
    for(float d = firstDepth; d < lastDepth;)
    {
        _depths.push_back(d);
        float finest = lastDepth - firstDepth;
        for(const auto& cand : dephtsPerTc)
        {
            const int k = indexOfNearestSorted(cand, d);
            if(k >= 0 && (std::size_t)k + 1 < cand.size())
                finest = std::min(finest, std::fabs(cand[k] - cand[k + 1]));
        }
        d += finest * scaleFactor;
    }
}

void SgmDepthList::exportTxtFiles(const std::vector<std::vector<float>>& dephtsPerTc) const
{
    const std::string stem = _mp.getDepthMapsFolder() + std::to_string(_mp.getViewId(_tile.rc)) + "_";
    const std::string tail = "_" + std::to_string(_tile.roi.x.begin) + "_" + std::to_string(_tile.roi.y.begin) + ".txt";
    auto dump = [&](const std::string& name, auto&& writeLines) {
        if(FILE* f = std::fopen((stem + name + tail).c_str(), "w"))
        {
            writeLines(f);
            std::fclose(f);
        }
    };
    dump("depthsTcLimits", [&](FILE* f) { for(const Pixel& l : _depthsTcLimits) std::fprintf(f, "%i %i\n", l.x, l.y); });
    dump("depths", [&](FILE* f) { for(const float d : _depths) std::fprintf(f, "%f\n", d); });
    for(std::size_t c = 0; c < dephtsPerTc.size(); ++c)
        dump("depths_tc_" + std::to_string(_mp.getViewId(_tile.sgmTCams.at(c))), [&](FILE* f) { for(const float d : dephtsPerTc[c]) std::fprintf(f, "%f\n", d); });
}

} // namespace avdm_host
