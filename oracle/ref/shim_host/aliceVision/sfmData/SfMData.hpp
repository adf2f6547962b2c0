// oracle/_ref (host): STAND-IN for aliceVision/sfmData/SfMData.hpp — the part of the scene model the depth-list and T-camera-selection
// code reads: landmarks (3-D point + observations per view id), views with their pose and pinhole intrinsic.  Containers ordered like
// the reference's (Landmarks = std::map<IndexT, Landmark>, SfMData.hpp:39; Observations = stl::flat_map, ordered by view id,
// Observation.hpp:65).  camera::angleBetweenRays is RESTATED here (camera/IntrinsicBase.hpp:475-517 is Eigen code over the camera
// model: for a pinhole without distortion the ray is R^T * normalize((x - c) / f, 1), normalised; the angle is
// degrees(acos(clamp(dot / (|r1| |r2|), -1 + 1e-8, 1 - 1e-8)))) — that one function stays unpinned.  Test infrastructure only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>

#include <aliceVision/numeric/standin_vec.hpp>

namespace aliceVision {
using IndexT = uint32_t;
namespace geometry {
struct Pose3
{
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; // row-major world -> camera rotation
};
} // namespace geometry
namespace camera {
struct IntrinsicBase
{
    double fx = 1, fy = 1, cx = 0, cy = 0;
};
inline void applyIntrinsicExtrinsic(const geometry::Pose3& pose, const IntrinsicBase* k, const Vec2& x, double out[3])
{
    double c[3] = {(x.v[0] - k->cx) / k->fx, (x.v[1] - k->cy) / k->fy, 1.0};
    const double n = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    c[0] /= n, c[1] /= n, c[2] /= n;
    for(int i = 0; i < 3; ++i) // R^T * c
        out[i] = pose.R[i] * c[0] + pose.R[3 + i] * c[1] + pose.R[6 + i] * c[2];
    const double m = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]);
    out[0] /= m, out[1] /= m, out[2] /= m;
}
inline double angleBetweenRays(const geometry::Pose3& pose1, const IntrinsicBase* intrinsic1, const geometry::Pose3& pose2, const IntrinsicBase* intrinsic2,
                               const Vec2& x1, const Vec2& x2)
{
    double r1[3], r2[3];
    applyIntrinsicExtrinsic(pose1, intrinsic1, x1, r1);
    applyIntrinsicExtrinsic(pose2, intrinsic2, x2, r2);
    const double mag = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]) * std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    const double dotAngle = r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2];
    const double v = std::max(-1.0 + 1.e-8, std::min(dotAngle / mag, 1.0 - 1.e-8));
    return std::acos(v) * 180.0 / 3.14159265358979323846;
}
} // namespace camera
namespace sfmData {
class View
{
  public:
    IndexT intrinsicId = 0;
    geometry::Pose3 pose;
    IndexT getIntrinsicId() const { return intrinsicId; }
};
struct CameraPose
{
    geometry::Pose3 transform;
    const geometry::Pose3& getTransform() const { return transform; }
};
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
using Views = std::map<IndexT, std::shared_ptr<View>>;
class SfMData
{
  public:
    Landmarks landmarks;
    Views views;
    std::map<IndexT, camera::IntrinsicBase> intrinsics;
    const Landmarks& getLandmarks() const { return landmarks; }
    const Views& getViews() const { return views; }
    CameraPose getPose(const View& v) const { return CameraPose{v.pose}; }
    const camera::IntrinsicBase* getIntrinsicPtr(IndexT id) const { return &intrinsics.at(id); }
};
} // namespace sfmData
} // namespace aliceVision
