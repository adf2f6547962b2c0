// SgmDepthList.hpp — the CPU stage that chooses the fronto-parallel depth planes of one R-camera tile and, per T camera,
// the sub-range of planes to sweep.  Restates depthMap/SgmDepthList.{hpp,cpp} of the reference.
#pragma once

#include "MultiViewParams.hpp"
#include "params.hpp"

#include <vector>

namespace avdm_host {

// SgmDepthList.cpp:25-42
int indexOfNearestSorted(const std::vector<float>& in_vector, const float value);

class SgmDepthList
{
  public:
    SgmDepthList(const MultiViewParams& mp, const SgmParams& sgmParams, const Tile& tile) : _mp(mp), _sgmParams(sgmParams), _tile(tile) {}

    const std::vector<float>& getDepths() const { return _depths; }
    const std::vector<Pixel>& getDepthsTcLimits() const { return _depthsTcLimits; } // (first plane index, number of planes) per T camera
    std::pair<float, float> getMinMaxDepths() const { return {_depths.front(), _depths.back()}; }

    void computeListRc();                            // :48-192
    void removeTcWithNoDepth(Tile& tile);            // :194-221
    void logRcTcDepthInformation() const;            // :223-248
    void checkStartingAndStoppingDepth() const;      // :250-275

  private:
    void getMinMaxMidNbDepthFromSfM(float& out_min, float& out_max, float& out_mid, std::size_t& out_nbDepths) const; // :277-345
    void getRcTcDepthRangeFromSfM(int tc, double& out_zmin, double& out_zmax) const;                                  // :347-415
    void computeRcTcDepths(int tc, float midDepth, std::vector<float>& out_depths) const;                             // :417-545
    void computePixelSizeDepths(float minObsDepth, float midObsDepth, float maxObsDepth, std::vector<float>& out_depths) const; // :547-625
    void computeRcDepthList(float firstDepth, float lastDepth, float scaleFactor, const std::vector<std::vector<float>>& dephtsPerTc); // :627-660
    void exportTxtFiles(const std::vector<std::vector<float>>& dephtsPerTc) const;                                    // :662-700

    const MultiViewParams& _mp;
    const SgmParams& _sgmParams;
    const Tile& _tile;
    std::vector<float> _depths;
    std::vector<Pixel> _depthsTcLimits;
};

} // namespace avdm_host
