// oracle/_ref (host): STAND-IN for Boost.Accumulators (third party, absent here) — the one statistic the depth-list code uses,
// tail_quantile<left | right> over a tail cache, with the call syntax of the library.  This part stays UNPINNED (it restates Boost's
// published algorithm, boost/accumulators/statistics/tail_quantile.hpp: n = ceil(count * p) for the left tail, ceil(count * (1 - p)) for
// the right tail; the n-th element of the sorted cached tail if n < tail size, else quiet NaN), exactly like oracle/host_oracle.py does.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <limits>
#include <vector>

namespace boost { namespace accumulators {
struct left {};
struct right {};
struct cache_size_arg { std::size_t n; };
struct cache_size_keyword { cache_size_arg operator=(std::size_t n) const { return cache_size_arg{n}; } };
struct probability_arg { double p; };
struct probability_keyword { probability_arg operator=(double p) const { return probability_arg{p}; } };
static const probability_keyword quantile_probability{};
namespace tag {
template <class LR> struct tail { static const cache_size_keyword cache_size; };
template <class LR> const cache_size_keyword tail<LR>::cache_size{};
template <class LR> struct tail_quantile { using side = LR; };
} // namespace tag
template <class Stat> struct stats { using side = typename Stat::side; };
template <class T, class Stats>
class accumulator_set
{
  public:
    explicit accumulator_set(cache_size_arg c) : _cache(c.n) {}
    void operator()(T v)
    {
        ++_count;
        _tail.push_back(v);
    }
    T quantile(double p) const
    {
        constexpr bool isLeft = std::is_same<typename Stats::side, left>::value;
        std::vector<T> t(_tail);
        if(isLeft)
            std::sort(t.begin(), t.end());
        else
            std::sort(t.begin(), t.end(), std::greater<T>());
        if(t.size() > _cache)
            t.resize(_cache);
        const std::size_t n = static_cast<std::size_t>(std::ceil(_count * (isLeft ? p : 1. - p)));
        if(n < t.size() && n > 0)
            return t[n - 1];
        return std::numeric_limits<T>::quiet_NaN();
    }

  private:
    std::size_t _cache, _count = 0;
    std::vector<T> _tail;
};
template <class T, class S>
T quantile(const accumulator_set<T, S>& acc, probability_arg p) { return acc.quantile(p.p); }
}} // namespace boost::accumulators
