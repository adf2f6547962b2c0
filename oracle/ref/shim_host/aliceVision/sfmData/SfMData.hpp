// oracle/_ref (host): STAND-IN for aliceVision/sfmData/SfMData.hpp — the part of the scene model the depth-list code reads: landmarks
// (3-D point + observations per view id).  Containers ordered like the reference's (Landmarks = std::map<IndexT, Landmark>,
// SfMData.hpp:39; Observations = stl::flat_map, ordered by view id, Observation.hpp:65).  Test infrastructure only.
#pragma once
#include <cstdint>
#include <map>

namespace aliceVision {
using IndexT = uint32_t;
struct Vec2
{
    double v[2];
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double operator()(int i) const { return v[i]; }
};
struct Vec3
{
    double v[3];
    double operator()(int i) const { return v[i]; }
};
namespace sfmData {
class Observation
{
  public:
    Vec2 coordinates;
    const Vec2& getCoordinates() const { return coordinates; }
    double getX() const { return coordinates.v[0]; }
    double getY() const { return coordinates.v[1]; }
};
using Observations = std::map<IndexT, Observation>;
struct Landmark
{
    Vec3 X;
    Observations observations;
    const Observations& getObservations() const { return observations; }
};
using Landmarks = std::map<IndexT, Landmark>;
class SfMData
{
  public:
    Landmarks landmarks;
    const Landmarks& getLandmarks() const { return landmarks; }
};
} // namespace sfmData
} // namespace aliceVision
