// SgmDepthList.cpp — see SgmDepthList.hpp.  float / double roles are kept as in the reference (depths are float, geometry is
// double) because the plane list feeds the kernels and has to be reproducible.
#include "SgmDepthList.hpp"

#include "log.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <limits>
#include <sstream>

namespace avdm_host {

int indexOfNearestSorted(const std::vector<float>& in_vector, const float value)
{
    auto it = std::lower_bound(in_vector.begin(), in_vector.end(), value);
    if(it == in_vector.end())
        return -1;
    if(it != in_vector.begin())
    {
        const auto prevIt = std::prev(it);
        it = (value - *prevIt) < (*it - value) ? prevIt : it;
    }
    return (int)std::distance(in_vector.begin(), it);
}

namespace {

// boost::accumulators tail_quantile with a tail cache of 1000 samples (boost/accumulators/statistics/tail_quantile.hpp,
// not part of the reference tree; published behaviour): the cache keeps the `cacheSize` most extreme samples sorted from the
// extreme inwards; quantile(p) = cache[ceil(count * (left ? p : 1 - p)) - 1] if that index is inside the cache, else NaN.
struct TailQuantile
{
    bool left;
    std::size_t cacheSize;
    std::size_t count = 0;
    std::vector<float> tail; // sorted: ascending for the left tail, descending for the right tail
    TailQuantile(bool l, std::size_t c) : left(l), cacheSize(c) {}
    void operator()(float v)
    {
        ++count;
        auto pos = left ? std::upper_bound(tail.begin(), tail.end(), v) : std::upper_bound(tail.begin(), tail.end(), v, std::greater<float>());
        if(tail.size() < cacheSize)
            tail.insert(pos, v);
        else if(pos != tail.end())
        {
            tail.insert(pos, v);
            tail.pop_back();
        }
    }
    float quantile(double probability) const
    {
        const std::size_t n = static_cast<std::size_t>(std::ceil(count * (left ? probability : 1. - probability)));
        if(n < tail.size())
            return n == 0 ? std::numeric_limits<float>::quiet_NaN() : tail[n - 1];
        return std::numeric_limits<float>::quiet_NaN();
    }
};

} // namespace

void SgmDepthList::computeListRc()
{
    AVDM_LOG_DEBUG(_tile << "Compute SGM depths list.");
    _depths.clear();
    _depthsTcLimits.clear();

    std::size_t nbObsDepths;
    float minObsDepth, maxObsDepth, midObsDepth;
    getMinMaxMidNbDepthFromSfM(minObsDepth, maxObsDepth, midObsDepth, nbObsDepths);
    if(nbObsDepths < 2)
    {
        AVDM_LOG_INFO(_tile << "Cannot get min/max/middle depth from SfM.");
        return;
    }

    std::vector<std::vector<float>> depthsPerTc(_tile.sgmTCams.size());
    for(std::size_t c = 0; c < _tile.sgmTCams.size(); ++c)
    {
        std::vector<float>& tcDepths = depthsPerTc.at(c);
        computeRcTcDepths(_tile.sgmTCams.at(c), (nbObsDepths < 10) ? -1 : midObsDepth, tcDepths);
        if(tcDepths.size() < 10)
        {
            AVDM_LOG_DEBUG(_tile << "Not enough valid samples over the epipolar line. Compute depth list from R camera pixel size.");
            tcDepths.clear();
            computePixelSizeDepths(minObsDepth, midObsDepth, maxObsDepth * (float)_sgmParams.prematchingMaxDepthScale, tcDepths);
        }
    }

    float minDepthAll = std::numeric_limits<float>::max();
    float maxDepthAll = std::numeric_limits<float>::min();
    for(const std::vector<float>& tcDepths : depthsPerTc)
        for(const float depth : tcDepths)
        {
            minDepthAll = std::min(minDepthAll, depth);
            maxDepthAll = std::max(maxDepthAll, depth);
        }
    if(minDepthAll > maxDepthAll)
    {
        AVDM_LOG_INFO(_tile << "No depths found.");
        return;
    }
    AVDM_LOG_DEBUG(_tile << "Depth candidates from seeds for R camera:" << std::endl
                         << "\t- nb observations: " << nbObsDepths << std::endl
                         << "\t- all depth range: [" << minDepthAll << "-" << maxDepthAll << "]" << std::endl
                         << "\t- sfm depth range: [" << minObsDepth << "-" << maxObsDepth << "]");

    float firstDepth = minDepthAll;
    float lastDepth = maxDepthAll;
    if(_sgmParams.useSfmSeeds && !_mp.getInputSfMData().landmarks.empty() && nbObsDepths > 10)
    {
        const float margin = _sgmParams.seedsRangeInflate * (maxObsDepth - minObsDepth);
        firstDepth = std::max(0.f, minObsDepth - margin);
        lastDepth = maxObsDepth + margin;
        if(maxDepthAll < firstDepth || minDepthAll > lastDepth)
        {
            // no intersection: keep the landmark range as is
        }
        else
        {
            firstDepth = std::max(minDepthAll, firstDepth);
            lastDepth = std::min(maxDepthAll, lastDepth);
        }
        AVDM_LOG_DEBUG(_tile << "Final depth range (intersection: frustums / landmarks with margin): [" << firstDepth << "-" << lastDepth << "]");
    }

    computeRcDepthList(firstDepth, lastDepth, (_sgmParams.stepZ > 0.0f ? _sgmParams.stepZ : 1.0f), depthsPerTc);

    if(_sgmParams.maxDepths > 0 && (int)_depths.size() > _sgmParams.maxDepths)
    {
        const float scaleFactor = float(_depths.size()) / float(_sgmParams.maxDepths);
        AVDM_LOG_DEBUG(_tile << "Too many values in R camera depth list, filter out with scale factor:" << std::endl
                             << "\t- nb depths: " << _depths.size() << std::endl
                             << "\t- max depths: " << _sgmParams.maxDepths << std::endl
                             << "\t- scale factor to apply: " << scaleFactor);
        computeRcDepthList(firstDepth, lastDepth, scaleFactor, depthsPerTc);
        if((int)_depths.size() > _sgmParams.maxDepths)
            _depths.resize(_sgmParams.maxDepths);
    }
    AVDM_LOG_DEBUG(_tile << "Final depth range for R camera:" << std::endl
                         << "\t- nb selected depths: " << _depths.size() << std::endl
                         << "\t- selected depth range: [" << firstDepth << "-" << lastDepth << "]");

    _depthsTcLimits.resize(_tile.sgmTCams.size());
    for(std::size_t c = 0; c < _tile.sgmTCams.size(); ++c)
    {
        if(depthsPerTc.empty())
        {
            _depthsTcLimits[c] = Pixel(-1, -1);
            continue;
        }
        const float d1 = depthsPerTc.at(c).front();
        const float d2 = depthsPerTc.at(c).back();
        int id1 = indexOfNearestSorted(_depths, d1);
        int id2 = indexOfNearestSorted(_depths, d2);
        if(id1 == -1)
            id1 = 0;
        if(id2 == -1)
            id2 = (int)_depths.size() - 1;
        _depthsTcLimits[c] = Pixel(id1, id2 - id1 + 1);
    }
    if(_sgmParams.exportDepthsTxtFiles)
        exportTxtFiles(depthsPerTc);
    AVDM_LOG_DEBUG(_tile << "Compute SGM depths list done.");
}

void SgmDepthList::removeTcWithNoDepth(Tile& tile)
{
    assert(tile.rc == _tile.rc);
    std::vector<int> out_tCams;
    std::vector<Pixel> out_depthsTcLimits;
    for(size_t c = 0; c < tile.sgmTCams.size(); ++c)
    {
        const Pixel& tcLimits = _depthsTcLimits.at(c);
        const int tc = tile.sgmTCams.at(c);
        if(tcLimits.x != -1 && tcLimits.y != -1)
        {
            out_tCams.push_back(tc);
            out_depthsTcLimits.push_back(tcLimits);
        }
        else
            AVDM_LOG_INFO(_tile << "Remove T camera (tc: " << tc << ", view id: " << _mp.getViewId(tc) << ") no depth found.");
    }
    std::swap(tile.sgmTCams, out_tCams);
    std::swap(_depthsTcLimits, out_depthsTcLimits);
}

void SgmDepthList::logRcTcDepthInformation() const
{
    std::ostringstream ostr;
    ostr << "Camera / Depth information: " << std::endl
         << "\t- R camera:" << std::endl
         << "\t   - id: " << _tile.rc << std::endl
         << "\t   - view id: " << _mp.getViewId(_tile.rc) << std::endl
         << "\t   - depth planes: " << _depths.size() << std::endl
         << "\t   - depths range: [" << _depths[0] << "-" << _depths[_depths.size() - 1] << "]" << std::endl
         << "\t- T cameras:" << std::endl;
    for(std::size_t c = 0; c < _tile.sgmTCams.size(); ++c)
        ostr << "\t   - T camera (" << (c + 1) << "/" << _tile.sgmTCams.size() << "):" << std::endl
             << "\t      - id: " << _tile.sgmTCams.at(c) << std::endl
             << "\t      - view id: " << _mp.getViewId(_tile.sgmTCams.at(c)) << std::endl
             << "\t      - depth planes: " << _depthsTcLimits[c].y << std::endl
             << "\t      - depths range: [" << _depths[_depthsTcLimits[c].x] << "-" << _depths[_depthsTcLimits[c].x + _depthsTcLimits[c].y - 1] << "]"
             << std::endl
             << "\t      - depth indexes range: [" << _depthsTcLimits[c].x << "-" << _depthsTcLimits[c].x + _depthsTcLimits[c].y << "]" << std::endl;
    AVDM_LOG_INFO(_tile << ostr.str());
}

void SgmDepthList::checkStartingAndStoppingDepth() const
{
    // the reference only asserts here (compiled out in release builds): starting index 0, stopping index <= number of planes
    if(_depthsTcLimits.empty())
        return;
    int startingDepth = std::numeric_limits<int>::max(), stoppingDepth = 0;
    for(const Pixel& l : _depthsTcLimits)
    {
        startingDepth = std::min(startingDepth, l.x);
        stoppingDepth = std::max(stoppingDepth, l.x + l.y);
    }
    if(startingDepth != 0 || (int)_depths.size() < stoppingDepth)
        AVDM_LOG_DEBUG(_tile << "Depth limits: starting depth index " << startingDepth << ", stopping depth index " << stoppingDepth << " / " << _depths.size());
}

void SgmDepthList::getMinMaxMidNbDepthFromSfM(float& out_min, float& out_max, float& out_mid, std::size_t& out_nbDepths) const
{
    const std::size_t cacheSize = 1000;
    TailQuantile accDistanceMin(true, cacheSize), accDistanceMax(false, cacheSize);

    const IndexT viewId = _mp.getViewId(_tile.rc);
    const ROI fullsizeRoi = upscaleROI(_tile.roi, (float)_mp.getProcessDownscale());

    const Point3d planeP = _mp.CArr[_tile.rc];
    const Point3d planeN = (_mp.iRArr[_tile.rc] * Point3d(0.0, 0.0, 1.0)).normalize();

    Point3d midDepthPoint;
    out_nbDepths = 0;
    for(const auto& landmarkPair : _mp.getInputSfMData().landmarks)
    {
        const Landmark& landmark = landmarkPair.second;
        const auto it = landmark.observations.find(viewId);
        if(it == landmark.observations.end())
            continue;
        if(!_sgmParams.depthListPerTile || fullsizeRoi.contains((unsigned int)it->second.x, (unsigned int)it->second.y))
        {
            const float distance = static_cast<float>(pointPlaneDistance(landmark.X, planeP, planeN));
            accDistanceMin(distance);
            accDistanceMax(distance);
            midDepthPoint = midDepthPoint + landmark.X;
            ++out_nbDepths;
        }
    }
    if(out_nbDepths > 0)
    {
        out_min = accDistanceMin.quantile(1.0 - _sgmParams.seedsRangePercentile);
        out_max = accDistanceMax.quantile(_sgmParams.seedsRangePercentile);
        midDepthPoint = midDepthPoint / static_cast<float>(out_nbDepths);
        out_mid = (float)pointPlaneDistance(midDepthPoint, planeP, planeN);
    }
    else
    {
        out_min = 0.f;
        out_max = 0.f;
        out_mid = 0.f;
    }
    AVDM_LOG_DEBUG(_tile << "Compute min/max/mid/nb observation depth from SfM for R camera:" << std::endl
                         << "\t- view id: " << viewId << std::endl
                         << "\t- min depth: " << out_min << std::endl
                         << "\t- max depth: " << out_max << std::endl
                         << "\t- mid depth: " << out_mid << std::endl
                         << "\t- nb depth: " << out_nbDepths << std::endl
                         << "\t- percentile: " << _sgmParams.seedsRangePercentile);
}

void SgmDepthList::getRcTcDepthRangeFromSfM(int tc, double& out_zmin, double& out_zmax) const
{
    const IndexT rcViewId = _mp.getViewId(_tile.rc);
    const IndexT tcViewId = _mp.getViewId(tc);
    const ROI fullsizeRoi = upscaleROI(_tile.roi, (float)_mp.getProcessDownscale());
    const Point3d planeP = _mp.CArr[_tile.rc];
    const Point3d planeN = (_mp.iRArr[_tile.rc] * Point3d(0.0, 0.0, 1.0)).normalize();

    out_zmin = std::numeric_limits<double>::max();
    out_zmax = std::numeric_limits<double>::min();
    for(const auto& landmarkPair : _mp.getInputSfMData().landmarks)
    {
        const Landmark& landmark = landmarkPair.second;
        if(landmark.observations.find(tcViewId) == landmark.observations.end())
            continue;
        const auto it = landmark.observations.find(rcViewId);
        if(it == landmark.observations.end())
            continue;
        if(!_sgmParams.depthListPerTile || fullsizeRoi.contains((unsigned int)it->second.x, (unsigned int)it->second.y))
        {
            const double depth = pointPlaneDistance(landmark.X, planeP, planeN);
            out_zmin = std::min(out_zmin, depth);
            out_zmax = std::max(out_zmax, depth);
        }
    }
    if(out_zmin > out_zmax)
        AVDM_THROW_ERROR(_tile << "Cannot compute min/max depth from common Rc/Tc SfM observations." << std::endl
                               << "No common observations found (tc view id: " << tcViewId << ").");
    AVDM_LOG_DEBUG(_tile << "Compute min/max depth from common Rc/Tc SfM observations:" << std::endl
                         << "\t- rc: " << _tile.rc << " (view id: " << rcViewId << ")" << std::endl
                         << "\t- tc: " << tc << " (view id: " << tcViewId << ")" << std::endl
                         << "\t- min depth: " << out_zmin << std::endl
                         << "\t- max depth: " << out_zmax);
}

void SgmDepthList::computeRcTcDepths(int tc, float midDepth, std::vector<float>& out_depths) const
{
    const Point3d rcplaneP = _mp.CArr[_tile.rc];
    const Point3d rcplaneN = (_mp.iRArr[_tile.rc] * Point3d(0.0, 0.0, 1.0)).normalize();

    const Point2d roiCenter((_tile.roi.x.begin + (_tile.roi.width() * 0.5)), _tile.roi.y.begin + (_tile.roi.height() * 0.5));
    const Point2d principalPoint(_mp.getWidth(_tile.rc) * 0.5, _mp.getHeight(_tile.rc) * 0.5);
    const Point2d referencePoint = (!_sgmParams.depthListPerTile) ? principalPoint : roiCenter;

    Point2d tcMidDepthPoint;
    Point2d tcFromPoint, tcToPoint; // stay (0,0) when the epipolar line misses the image, like the reference's default Point2d
    {
        const Matrix3x4& rP = _mp.camArr[_tile.rc];
        const Matrix3x4& tP = _mp.camArr[tc];
        Point3d rC;
        Matrix3x3 rR, riR, rK, riK, riP;
        _mp.decomposeProjectionMatrix(rC, rR, riR, rK, riK, riP, rP);
        _mp.getPixelFor3DPoint(&tcMidDepthPoint, ((riP * referencePoint) * midDepth) + rC, tP);

        double zmin, zmax;
        getRcTcDepthRangeFromSfM(tc, zmin, zmax);
        Point2d tarpix1, tarpix2;
        _mp.getPixelFor3DPoint(&tarpix1, ((riP * referencePoint) * zmin) + rC, tP);
        _mp.getPixelFor3DPoint(&tarpix2, ((riP * referencePoint) * zmax) + rC, tP);
        get2dLineImageIntersection(&tcFromPoint, &tcToPoint, tarpix1, tarpix2, _mp, tc);
    }

    const int nbSegmentPoints = static_cast<int>((tcToPoint - tcFromPoint).size());
    const int nbSegmentPointsAtSgmScale = nbSegmentPoints / _sgmParams.scale;
    const Point2d pixelVect = (tcToPoint - tcFromPoint).normalize() * std::max(1.0, double(_sgmParams.scale));

    int depthDirection = 1;
    {
        Point3d p;
        if(!triangulateMatch(p, referencePoint, tcMidDepthPoint, _tile.rc, tc, _mp))
            return;
        const float depth = (float)orientedPointPlaneDistance(p, rcplaneP, rcplaneN);
        if(!triangulateMatch(p, referencePoint, tcMidDepthPoint + pixelVect, _tile.rc, tc, _mp))
            return;
        const float depthP1 = (float)orientedPointPlaneDistance(p, rcplaneP, rcplaneN);
        if(depth > depthP1)
            depthDirection = -1;
    }

    out_depths.reserve(std::max(nbSegmentPointsAtSgmScale, 0));
    const Point3d refVect = _mp.iCamArr[_tile.rc] * referencePoint;
    float previousDepth = -1.0f;

    for(int i = 0; i < nbSegmentPointsAtSgmScale; ++i)
    {
        const Point2d tcPoint = ((depthDirection > 0) ? tcFromPoint : tcToPoint) + (pixelVect * double(i) * double(depthDirection));
        if(!_mp.isPixelInImage(tcPoint, tc))
            continue;
        const Point3d tarVect = _mp.iCamArr[tc] * tcPoint;
        const float refTarVectAngle = (float)angleBetwV1andV2(refVect, tarVect);
        if(refTarVectAngle < _mp.getMinViewAngle() || refTarVectAngle > _mp.getMaxViewAngle())
            continue;
        Point3d p;
        if(!triangulateMatch(p, referencePoint, tcPoint, _tile.rc, tc, _mp))
            continue;
        const float depth = float(orientedPointPlaneDistance(p, rcplaneP, rcplaneN));
        if((depth > 0.0f) && (depth > previousDepth))
        {
            out_depths.push_back(depth);
            previousDepth = depth + std::numeric_limits<float>::epsilon();
        }
    }
    out_depths.shrink_to_fit();

    AVDM_LOG_DEBUG(_tile << "Find depths over the epipolar line segment between R and T cameras:" << std::endl
                         << "\t- rc: " << _tile.rc << "(view id: " << _mp.getViewId(_tile.rc) << ")" << std::endl
                         << "\t- tc: " << tc << "(view id: " << _mp.getViewId(tc) << ")" << std::endl
                         << "\t- # points of the epipolar segment: " << nbSegmentPoints << std::endl
                         << "\t- # points of the epipolar segment at SGM scale: " << nbSegmentPointsAtSgmScale << std::endl
                         << "\t- # depths to use: " << out_depths.size());
    if(!out_depths.empty())
        AVDM_LOG_DEBUG(_tile << "Depth to use range [" << out_depths.front() << "-" << out_depths.back() << "]" << std::endl);
}

void SgmDepthList::computePixelSizeDepths(float minObsDepth, float midObsDepth, float maxObsDepth, std::vector<float>& out_depths) const
{
    const int rcDepthsCompStep = 6;
    const int maxDepthsHalf = 1024;
    const float d = float(_sgmParams.scale) * float(rcDepthsCompStep);

    const Point3d planeP = _mp.CArr[_tile.rc];
    const Point3d planeN = (_mp.iRArr[_tile.rc] * Point3d(0.0, 0.0, 1.0)).normalize();

    int ndepthsMidMax = 0;
    float maxdepth = midObsDepth;
    while((maxdepth < maxObsDepth) && (ndepthsMidMax < maxDepthsHalf))
    {
        const Point3d p = planeP + planeN * maxdepth;
        const float pixSize = (float)_mp.getCamPixelSize(p, _tile.rc, d);
        maxdepth += pixSize;
        ndepthsMidMax++;
    }
    int ndepthsMidMin = 0;
    float mindepth = midObsDepth;
    while((mindepth > minObsDepth) && (ndepthsMidMin < maxDepthsHalf * 2 - ndepthsMidMax))
    {
        const Point3d p = planeP + planeN * mindepth;
        const float pixSize = (float)_mp.getCamPixelSize(p, _tile.rc, d);
        mindepth -= pixSize;
        ndepthsMidMin++;
    }
    float depth = mindepth;
    float pixSize = 1.0f;
    int ndepths = 0;
    while((depth < maxdepth) && (pixSize > 0.0f) && (ndepths < 2 * maxDepthsHalf))
    {
        out_depths.push_back(depth);
        const Point3d p = planeP + planeN * depth;
        pixSize = (float)_mp.getCamPixelSize(p, _tile.rc, d);
        depth += pixSize;
        ndepths++;
    }
    for(size_t i = 0; i + 1 < out_depths.size(); i++)
        if(out_depths[i] >= out_depths[i + 1])
            throw std::runtime_error("getDepthsByPixelSize not asc.");
}

void SgmDepthList::computeRcDepthList(float firstDepth, float lastDepth, float scaleFactor, const std::vector<std::vector<float>>& dephtsPerTc)
{
    _depths.clear();
    float depth = firstDepth;
    while(depth < lastDepth)
    {
        _depths.push_back(depth);
        float minTcStep = lastDepth - firstDepth;
        for(const std::vector<float>& tcDepths : dephtsPerTc)
        {
            const int id = indexOfNearestSorted(tcDepths, depth);
            // `id >= tcDepths.size() - 1` is an int / size_t comparison in the reference: id = -1 converts to SIZE_MAX and is skipped too
            if(id < 0 || (size_t)id >= tcDepths.size() - 1)
                continue;
            const float tcStep = std::fabs(tcDepths.at(id) - tcDepths.at(id + 1));
            minTcStep = std::min(minTcStep, tcStep);
        }
        depth += minTcStep * scaleFactor;
    }
}

void SgmDepthList::exportTxtFiles(const std::vector<std::vector<float>>& dephtsPerTc) const
{
    const std::string prefix(_mp.getDepthMapsFolder() + std::to_string(_mp.getViewId(_tile.rc)) + std::string("_"));
    const std::string suffix("_" + std::to_string(_tile.roi.x.begin) + "_" + std::to_string(_tile.roi.y.begin) + ".txt");
    if(FILE* f = std::fopen((prefix + "depthsTcLimits" + suffix).c_str(), "w"))
    {
        for(const Pixel& l : _depthsTcLimits)
            std::fprintf(f, "%i %i\n", l.x, l.y);
        std::fclose(f);
    }
    if(FILE* f = std::fopen((prefix + "depths" + suffix).c_str(), "w"))
    {
        for(const float dd : _depths)
            std::fprintf(f, "%f\n", dd);
        std::fclose(f);
    }
    for(size_t c = 0; c < dephtsPerTc.size(); ++c)
        if(FILE* f = std::fopen((prefix + "depths_tc_" + std::to_string(_mp.getViewId(_tile.sgmTCams.at(c))) + suffix).c_str(), "w"))
        {
            for(const float depth : dephtsPerTc.at(c))
                std::fprintf(f, "%f\n", depth);
            std::fclose(f);
        }
}

} // namespace avdm_host
