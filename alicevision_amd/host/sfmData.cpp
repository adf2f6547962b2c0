// sfmData.cpp — .sfm / .json scene reader (see sfmData.hpp for the reference lines restated).
#include "sfmData.hpp"

#include "alembic.hpp"

#include "json.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>

namespace avdm_host {

namespace {

struct Version
{
    int a = 0, b = 0, c = 0;
    bool operator<(const Version& o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); }
};

void loadVector(const JsonValue& arr, double* out, int n)
{
    if(arr.kind != JsonValue::Array)
        throw std::runtime_error("JSON: expected an array");
    for(int i = 0; i < n && i < (int)arr.items.size(); ++i)
        out[i] = arr.items[i].asDouble();
}

// radial_distortion::bisection_Radius_Solve (camera/DistortionRadial.hpp:26-46)
template <typename F>
double bisectionRadiusSolve(F functor, double r2, double epsilon = 1e-8)
{
    double lowerbound = r2, upbound = r2;
    while(functor(lowerbound) > r2)
        lowerbound /= 1.05;
    while(functor(upbound) < r2)
        upbound *= 1.05;
    while(epsilon < (upbound - lowerbound))
    {
        const double mid = .5 * (lowerbound + upbound);
        if(functor(mid) > r2)
            upbound = mid;
        else
            lowerbound = mid;
    }
    return .5 * (lowerbound + upbound);
}

} // namespace

// camera/DistortionRadial.cpp:45-53 (K1), :104-108,177-187 (K3)
Point2d Intrinsic::removeDistortion(const Point2d& p) const
{
    if(distortionType == "radialk1" && distortionParams.size() >= 1)
    {
        const double k1 = distortionParams[0];
        const double r2 = p.x * p.x + p.y * p.y;
        const double radius = (r2 == 0) ? 1. : std::sqrt(bisectionRadiusSolve([&](double x) { const double c = 1. + k1 * x; return x * c * c; }, r2) / r2);
        return p * radius;
    }
    if(distortionType == "radialk3" && distortionParams.size() >= 3)
    {
        const double k1 = distortionParams[0], k2 = distortionParams[1], k3 = distortionParams[2];
        const double r2 = p.x * p.x + p.y * p.y;
        const double radius =
          (r2 == 0) ? 1. : std::sqrt(bisectionRadiusSolve([&](double x) { const double c = 1. + x * (k1 + x * (k2 + x * k3)); return x * c * c; }, r2) / r2);
        return p * radius;
    }
    return p; // "none" (other models: see the warning printed by loadSfMData)
}

double angleBetweenRays(const Pose& pose1, const Intrinsic& intr1, const Pose& pose2, const Intrinsic& intr2, const Point2d& x1, const Point2d& x2)
{
    auto ray = [](const Pose& pose, const Intrinsic& intr, const Point2d& x) {
        const Point2d c = intr.removeDistortion(intr.ima2cam(x));
        const Point3d unit = Point3d(c.x, c.y, 1.0).normalize(); // Pinhole::toUnitSphere
        const Matrix3x3& R = pose.rotation;
        // R^T * unit
        const Point3d w(R(0, 0) * unit.x + R(1, 0) * unit.y + R(2, 0) * unit.z, R(0, 1) * unit.x + R(1, 1) * unit.y + R(2, 1) * unit.z,
                        R(0, 2) * unit.x + R(1, 2) * unit.y + R(2, 2) * unit.z);
        return w.normalize();
    };
    const Point3d r1 = ray(pose1, intr1, x1), r2 = ray(pose2, intr2, x2);
    const double mag = r1.size() * r2.size();
    const double c = std::min(std::max(dot(r1, r2) / mag, -1.0 + 1.e-8), 1.0 - 1.e-8);
    return std::acos(c) * 180.0 / M_PI;
}

bool ExposureSetting::hasShutter() const { return shutter > 0.0 && std::isnormal(shutter); }
bool ExposureSetting::hasFNumber() const { return fnumber > 0.0 && std::isnormal(fnumber); }

double ExposureSetting::getExposure() const
{
    // sfmData/ExposureSetting.hpp:37-112 with referenceISO = 100, referenceFNumber = 1
    const double referenceISO = 100.0, referenceFNumber = 1.0;
    if(!hasShutter() && !hasFNumber())
        return -1.0;
    const double sh = hasShutter() ? shutter : 1.0 / 200.0;
    const double fn = hasFNumber() ? fnumber : referenceFNumber;
    double iso2Aperture = 1.0;
    if(iso > 1e-6 && referenceISO > 1e-6)
        iso2Aperture = std::sqrt(iso / referenceISO);
    const double newFnumber = fn * iso2Aperture;
    const double expIncrease = (referenceFNumber / newFnumber) * (referenceFNumber / newFnumber);
    return sh * expIncrease;
}

namespace {
const std::string* findMetadata(const std::map<std::string, std::string>& md, const std::string& name)
{
    const auto it = md.find(name);
    if(it != md.end())
        return &it->second;
    std::string nameLower = name;
    std::transform(nameLower.begin(), nameLower.end(), nameLower.begin(), ::tolower);
    for(const auto& kv : md)
    {
        std::string key = kv.first;
        std::transform(key.begin(), key.end(), key.begin(), ::tolower);
        if(key.size() > name.size())
        {
            const size_t d = key.find_last_of("/:");
            if(d != std::string::npos)
                key = key.substr(d + 1);
        }
        if(key == nameLower)
            return &kv.second;
    }
    return nullptr;
}
const std::string* findMetadata(const std::map<std::string, std::string>& md, std::initializer_list<const char*> names)
{
    for(const char* n : names)
        if(const std::string* v = findMetadata(md, n))
            return v;
    return nullptr;
}
// ImageInfo::readRealNumber: "num/den" anywhere in the text, else std::stod; -1 on failure
double readRealNumber(const std::string& str)
{
    try
    {
        size_t i = 0;
        while(i < str.size())
        {
            // the first run of digits directly followed by '/' and a digit
            if(std::isdigit((unsigned char)str[i]))
            {
                size_t j = i;
                while(j < str.size() && std::isdigit((unsigned char)str[j]))
                    ++j;
                if(j + 1 < str.size() && str[j] == '/' && std::isdigit((unsigned char)str[j + 1]))
                {
                    size_t k = j + 1;
                    while(k < str.size() && std::isdigit((unsigned char)str[k]))
                        ++k;
                    // std::regex_search finds the leftmost match; with greedy digit runs that is this one, except that the numerator may
                    // start inside a longer digit run only at its beginning — which is where we are
                    const int num = std::stoi(str.substr(i, j - i)), den = std::stoi(str.substr(j + 1, k - j - 1));
                    return den != 0 ? double(num) / double(den) : 0.0;
                }
                i = j;
            }
            else
                ++i;
        }
        return std::stod(str);
    }
    catch(const std::exception&)
    {
        return -1.0;
    }
}
// ImageInfo::hasDigitMetadata(names, isPositive = true): the first of the names whose value parses as a number decides
bool hasDigitMetadata(const std::map<std::string, std::string>& md, std::initializer_list<const char*> names)
{
    for(const char* n : names)
    {
        const std::string* v = findMetadata(md, n);
        if(v == nullptr || v->empty())
            continue;
        try
        {
            return std::stod(*v) > 0.0;
        }
        catch(const std::exception&)
        {
        }
    }
    return false;
}
// ImageInfo::getDoubleMetadata(names): the first of the names that exists; -1 when none does or its value is empty
double getDoubleMetadata(const std::map<std::string, std::string>& md, std::initializer_list<const char*> names)
{
    const std::string* v = findMetadata(md, names);
    return v == nullptr || v->empty() ? -1.0 : readRealNumber(*v);
}
} // namespace

ExposureSetting cameraExposureSetting(const std::map<std::string, std::string>& md)
{
    ExposureSetting e;
    e.shutter = getDoubleMetadata(md, {"ExposureTime", "Shutter Speed Value"});
    // ImageInfo.hpp:213-226
    if(hasDigitMetadata(md, {"FNumber"}))
        e.fnumber = getDoubleMetadata(md, {"FNumber"});
    else if(hasDigitMetadata(md, {"ApertureValue", "Aperture Value"}))
        e.fnumber = std::pow(2.0, getDoubleMetadata(md, {"ApertureValue", "Aperture Value"}) / 2.0);
    e.iso = getDoubleMetadata(md, {"Exif:PhotographicSensitivity", "PhotographicSensitivity", "Photographic Sensitivity", "ISO"});
    return e;
}

double SfMData::medianCameraExposure() const
{
    std::vector<double> exposures; // distinct by value: ExposureSetting::operator== compares getExposure()
    for(const auto& kv : views)
    {
        const ExposureSetting ce = cameraExposureSetting(kv.second.metadata);
        if(!ce.isPartiallyDefined())
            continue;
        const double x = ce.getExposure();
        if(std::find(exposures.begin(), exposures.end(), x) == exposures.end())
            exposures.push_back(x);
    }
    if(exposures.empty())
        return -1.0;
    std::nth_element(exposures.begin(), exposures.begin() + exposures.size() / 2, exposures.end());
    return exposures[exposures.size() / 2];
}

void loadSfMData(SfMData& out, const std::string& filename)
{
    const size_t dot = filename.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : filename.substr(dot);
    if(ext == ".abc")
        return loadSfMDataAlembic(out, filename); // sfmDataIO::load dispatches on the extension (sfmDataIO.cpp:106-131)
    std::ifstream f(filename, std::ios::binary);
    if(!f)
        throw std::runtime_error("cannot open '" + filename + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    const JsonValue root = JsonParser(text).parse();
    if(root.kind != JsonValue::Object)
        throw std::runtime_error("SfMData: top-level JSON value is not an object");

    Version version;
    if(const JsonValue* v = root.find("version"))
    {
        double vv[3] = {0, 0, 0};
        loadVector(*v, vv, 3);
        version = {(int)vv[0], (int)vv[1], (int)vv[2]};
    }

    if(const JsonValue* intr = root.find("intrinsics"))
        for(const JsonValue& n : intr->items)
        {
            Intrinsic I;
            I.intrinsicId = (IndexT)n.at("intrinsicId").asUInt();
            std::string type = n.getString("type", "pinhole");
            std::transform(type.begin(), type.end(), type.begin(), ::tolower);
            if(version < Version{1, 2, 8})
            { // camera/cameraCommon.hpp:206-270 compatibilityStringToEnums
                I.isPinhole = type != "equidistant" && type != "equidistant_r3";
                I.distortionType = type == "radial1" ? "radialk1" : (type == "radial3" ? "radialk3" : (type == "pinhole" || type == "3deanamorphic4" ? "none" : type));
                I.type = I.isPinhole ? "pinhole" : "equidistant";
            }
            else
            {
                I.type = type;
                I.isPinhole = (type == "pinhole");
                I.distortionType = n.getString("distortionType", "none");
            }
            I.width = (int)n.at("width").asUInt();
            I.height = (int)n.at("height").asUInt();
            I.sensorWidth = n.getDouble("sensorWidth", 36.0);
            I.sensorHeight = n.getDouble("sensorHeight", 24.0);
            double pp[2] = {0, 0};
            if(const JsonValue* p = n.find("principalPoint"))
                loadVector(*p, pp, 2);
            if(version < Version{1, 2, 1})
            {
                pp[0] -= I.width / 2.0;
                pp[1] -= I.height / 2.0;
            }
            I.offsetX = pp[0];
            I.offsetY = pp[1];
            // focal length (jsonIO.cpp:302-346, IntrinsicScaleOffset.cpp:215-232)
            if(version < Version{1, 2, 0})
                I.scaleX = I.scaleY = n.getDouble("pxFocalLength", -1);
            else if(version < Version{1, 2, 2})
            {
                double fl[2] = {1, 1};
                loadVector(n.at("pxFocalLength"), fl, 2);
                I.scaleX = fl[0];
                I.scaleY = fl[1];
            }
            else if(version < Version{1, 2, 5})
            {
                const double fmm = n.getDouble("focalLength", 1.0), focalRatio = n.getDouble("pixelRatio", 1.0);
                I.scaleX = (fmm / I.sensorWidth) * double(I.width);
                I.scaleY = I.scaleX / focalRatio;
            }
            else
            {
                const double fmm = n.getDouble("focalLength", 1.0), par = n.getDouble("pixelRatio", 1.0);
                const double mm2px = double(I.width) / I.sensorWidth;
                if(version < Version{1, 2, 11})
                {
                    I.scaleX = fmm * mm2px;
                    I.scaleY = fmm * par * mm2px;
                }
                else
                {
                    I.scaleX = (fmm / par) * mm2px;
                    I.scaleY = fmm * mm2px;
                }
            }
            if(const JsonValue* dp = n.find("distortionParams"))
                for(const JsonValue& d : dp->items)
                    I.distortionParams.push_back(d.asDouble());
            const bool anyDisto = std::any_of(I.distortionParams.begin(), I.distortionParams.end(), [](double d) { return d != 0.0; });
            if(anyDisto && I.distortionType != "none" && I.distortionType != "radialk1" && I.distortionType != "radialk3")
                std::cerr << "[warning] intrinsic " << I.intrinsicId << ": distortion model '" << I.distortionType
                          << "' is not restated; observations are used as undistorted for the view-angle tests." << std::endl;
            out.intrinsics[I.intrinsicId] = I;
        }

    if(const JsonValue* views = root.find("views"))
        for(const JsonValue& n : views->items)
        {
            View v;
            v.viewId = (IndexT)n.getUInt("viewId", UndefinedIndexT);
            v.poseId = (IndexT)n.getUInt("poseId", UndefinedIndexT);
            v.intrinsicId = (IndexT)n.getUInt("intrinsicId", UndefinedIndexT);
            if(n.has("rigId"))
            { // jsonIO.cpp:84-91
                v.rigId = (IndexT)n.getUInt("rigId", UndefinedIndexT);
                v.subPoseId = (IndexT)n.getUInt("subPoseId", UndefinedIndexT);
            }
            if(const JsonValue* ind = n.find("isPoseIndependant"))
                v.independantPose = ind->asBool();
            v.path = n.getString("path", "");
            v.width = (int)n.getUInt("width", 0);
            v.height = (int)n.getUInt("height", 0);
            if(const JsonValue* md = n.find("metadata"))
                for(const auto& kv : md->members)
                    v.metadata[kv.first] = kv.second.text;
            out.views[v.viewId] = v;
        }

    if(const JsonValue* poses = root.find("poses"))
        for(const JsonValue& n : poses->items)
        {
            const IndexT poseId = (IndexT)n.at("poseId").asUInt();
            const JsonValue& tr = n.at("pose").at("transform");
            double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c[3] = {0, 0, 0};
            loadVector(tr.at("rotation"), r, 9);
            loadVector(tr.at("center"), c, 3);
            Pose p;
            // loadMatrix fills matrix(i) in Eigen's storage order, which is COLUMN-major for Mat3 (jsonIO.hpp:49-63)
            for(int col = 0; col < 3; ++col)
                for(int row = 0; row < 3; ++row)
                    p.rotation(row, col) = r[3 * col + row];
            p.center = Point3d(c[0], c[1], c[2]);
            out.poses[poseId] = p;
        }

    if(const JsonValue* rigs = root.find("rigs"))
        for(const JsonValue& n : rigs->items)
        { // jsonIO.cpp:472-489 loadRig
            Rig rig;
            if(const JsonValue* sps = n.find("subPoses"))
                for(const JsonValue& spn : sps->items)
                {
                    RigSubPose sp;
                    std::string status = spn.getString("status", "uninitialized");
                    std::transform(status.begin(), status.end(), status.begin(), ::tolower);
                    if(status != "uninitialized" && status != "estimated" && status != "constant")
                        throw std::runtime_error("SfMData: invalid rigSubPoseStatus '" + status + "'");
                    sp.initialized = status != "uninitialized";
                    const JsonValue& pn = spn.at("pose");
                    double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c[3] = {0, 0, 0};
                    loadVector(pn.at("rotation"), r, 9);
                    loadVector(pn.at("center"), c, 3);
                    for(int col = 0; col < 3; ++col)
                        for(int row = 0; row < 3; ++row)
                            sp.pose.rotation(row, col) = r[3 * col + row];
                    sp.pose.center = Point3d(c[0], c[1], c[2]);
                    rig.subPoses.push_back(sp);
                }
            out.rigs.emplace((IndexT)n.at("rigId").asUInt(), std::move(rig));
        }

    if(const JsonValue* st = root.find("structure"))
        for(const JsonValue& n : st->items)
        {
            const IndexT id = (IndexT)n.at("landmarkId").asUInt();
            Landmark L;
            double X[3] = {0, 0, 0};
            loadVector(n.at("X"), X, 3);
            L.X = Point3d(X[0], X[1], X[2]);
            if(const JsonValue* col = n.find("color"))
            {
                double c[3] = {255, 255, 255};
                loadVector(*col, c, 3);
                for(int k = 0; k < 3; ++k)
                    L.rgb[k] = (unsigned char)c[k];
            }
            if(const JsonValue* obs = n.find("observations"))
                for(const JsonValue& o : obs->items)
                {
                    Observation ob;
                    double x[2] = {0, 0};
                    if(const JsonValue* xv = o.find("x"))
                        loadVector(*xv, x, 2);
                    ob.x = x[0];
                    ob.y = x[1];
                    L.observations[(IndexT)o.at("observationId").asUInt()] = ob;
                }
            out.landmarks[id] = std::move(L);
        }
}

} // namespace avdm_host
