// json.hpp — minimal JSON DOM reader for AliceVision's .sfm / .json scene files.
// The reference reads them with boost::property_tree (sfmDataIO/jsonIO.cpp), which stores every scalar as text — its writer
// emits numbers as quoted strings ("viewId": "12345").  Scalars are therefore kept as text here too and converted on access,
// so quoted and bare numbers are both accepted.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace avdm_host {

struct JsonValue
{
    enum Kind { Null, Scalar, Array, Object } kind = Null;
    std::string text;                                         // Scalar (string, number, true/false)
    std::vector<JsonValue> items;                             // Array
    std::vector<std::pair<std::string, JsonValue>> members;   // Object, in file order

    const JsonValue* find(const std::string& key) const
    {
        for(const auto& kv : members)
            if(kv.first == key)
                return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    const JsonValue& at(const std::string& key) const
    {
        const JsonValue* v = find(key);
        if(!v)
            throw std::runtime_error("JSON: missing key '" + key + "'");
        return *v;
    }
    double asDouble() const { return std::strtod(text.c_str(), nullptr); }
    long long asInt() const { return std::strtoll(text.c_str(), nullptr, 10); }
    uint64_t asUInt() const { return std::strtoull(text.c_str(), nullptr, 10); }
    bool asBool() const { return text == "true" || text == "1"; }
    std::string getString(const std::string& key, const std::string& dflt) const
    {
        const JsonValue* v = find(key);
        return v ? v->text : dflt;
    }
    double getDouble(const std::string& key, double dflt) const
    {
        const JsonValue* v = find(key);
        return v ? v->asDouble() : dflt;
    }
    uint64_t getUInt(const std::string& key, uint64_t dflt) const
    {
        const JsonValue* v = find(key);
        return v ? v->asUInt() : dflt;
    }
};

class JsonParser
{
  public:
    explicit JsonParser(const std::string& s) : _s(s) {}
    JsonValue parse()
    {
        JsonValue v = value();
        ws();
        if(_p != _s.size())
            fail("trailing characters");
        return v;
    }

  private:
    const std::string& _s;
    size_t _p = 0;
    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error("JSON parse error at byte " + std::to_string(_p) + ": " + what); }
    void ws()
    {
        while(_p < _s.size() && (_s[_p] == ' ' || _s[_p] == '\n' || _s[_p] == '\t' || _s[_p] == '\r'))
            ++_p;
    }
    std::string str()
    {
        std::string out;
        ++_p; // opening quote
        while(_p < _s.size() && _s[_p] != '"')
        {
            char c = _s[_p++];
            if(c == '\\')
            {
                if(_p >= _s.size())
                    fail("bad escape");
                const char e = _s[_p++];
                switch(e)
                {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u':
                    {
                        if(_p + 4 > _s.size())
                            fail("bad \\u escape");
                        const unsigned cp = (unsigned)std::strtoul(_s.substr(_p, 4).c_str(), nullptr, 16);
                        _p += 4;
                        if(cp < 0x80)
                            out += (char)cp;
                        else if(cp < 0x800)
                        {
                            out += (char)(0xc0 | (cp >> 6));
                            out += (char)(0x80 | (cp & 0x3f));
                        }
                        else
                        {
                            out += (char)(0xe0 | (cp >> 12));
                            out += (char)(0x80 | ((cp >> 6) & 0x3f));
                            out += (char)(0x80 | (cp & 0x3f));
                        }
                        break;
                    }
                    default: out += e; // \" \\ \/
                }
            }
            else
                out += c;
        }
        if(_p >= _s.size())
            fail("unterminated string");
        ++_p;
        return out;
    }
    JsonValue value()
    {
        ws();
        if(_p >= _s.size())
            fail("unexpected end");
        JsonValue v;
        const char c = _s[_p];
        if(c == '{')
        {
            v.kind = JsonValue::Object;
            ++_p;
            ws();
            if(_p < _s.size() && _s[_p] == '}')
            {
                ++_p;
                return v;
            }
            for(;;)
            {
                ws();
                if(_p >= _s.size() || _s[_p] != '"')
                    fail("expected member name");
                std::string key = str();
                ws();
                if(_p >= _s.size() || _s[_p] != ':')
                    fail("expected ':'");
                ++_p;
                v.members.emplace_back(std::move(key), value());
                ws();
                if(_p < _s.size() && _s[_p] == ',')
                {
                    ++_p;
                    continue;
                }
                if(_p < _s.size() && _s[_p] == '}')
                {
                    ++_p;
                    return v;
                }
                fail("expected ',' or '}'");
            }
        }
        if(c == '[')
        {
            v.kind = JsonValue::Array;
            ++_p;
            ws();
            if(_p < _s.size() && _s[_p] == ']')
            {
                ++_p;
                return v;
            }
            for(;;)
            {
                v.items.push_back(value());
                ws();
                if(_p < _s.size() && _s[_p] == ',')
                {
                    ++_p;
                    continue;
                }
                if(_p < _s.size() && _s[_p] == ']')
                {
                    ++_p;
                    return v;
                }
                fail("expected ',' or ']'");
            }
        }
        if(c == '"')
        {
            v.kind = JsonValue::Scalar;
            v.text = str();
            return v;
        }
        // bare token: number, true, false, null
        const size_t b = _p;
        while(_p < _s.size() && _s[_p] != ',' && _s[_p] != '}' && _s[_p] != ']' && _s[_p] != ' ' && _s[_p] != '\n' && _s[_p] != '\t' && _s[_p] != '\r')
            ++_p;
        if(_p == b)
            fail("unexpected character");
        v.text = _s.substr(b, _p - b);
        v.kind = (v.text == "null") ? JsonValue::Null : JsonValue::Scalar;
        return v;
    }
};

} // namespace avdm_host
