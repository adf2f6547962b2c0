// cmdline.hpp — the small "--name value" command-line parser shared by the executables: what cmdline/cmdline.{hpp,cpp} + boost
// program_options give the reference's programs (required / optional options with textual defaults, `-x` short names, `--name=value`,
// booleans as values, the "Program called with the following parameters" echo, --help).
#pragma once

#include <cctype>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace avdm_host {

struct Option
{
    std::string name;      // long name
    char shortName;        // 0 if none
    bool required;
    std::string help;
    std::string dflt;      // textual default
    std::function<void(const std::string&)> set;
    bool hidden;
    bool multitoken = false; // boost's ->multitoken(): the option takes every following argument up to the next option
};

class CmdLine
{
  public:
    explicit CmdLine(const std::string& desc) : _desc(desc) {}
    template <typename T>
    void add(const std::string& name, T* target, const std::string& help, bool required = false, char shortName = 0, bool hidden = false)
    {
        Option o;
        o.name = name, o.shortName = shortName, o.required = required, o.help = help, o.hidden = hidden;
        std::ostringstream d;
        d << *target;
        o.dflt = d.str();
        o.set = [target, name](const std::string& v) { parse(name, v, target); };
        _options.push_back(o);
    }
    // po::value<std::vector<T>>()->multitoken(): tokens are collected as text, the caller parses them
    void addMultitoken(const std::string& name, std::vector<std::string>* target, const std::string& help)
    {
        Option o;
        o.name = name, o.shortName = 0, o.required = false, o.help = help, o.hidden = false, o.multitoken = true;
        o.dflt = "";
        o.set = [target](const std::string& v) { target->push_back(v); };
        _options.push_back(o);
    }
    // returns false when the program should stop (help or error; `error` tells which)
    bool execute(int argc, char** argv, bool& error)
    {
        error = false;
        std::map<std::string, bool> seen;
        for(int i = 1; i < argc; ++i)
        {
            std::string a = argv[i], value;
            bool hasValue = false;
            const Option* opt = nullptr;
            if(a == "--help" || a == "-h")
            {
                printHelp();
                return false;
            }
            if(a.rfind("--", 0) == 0)
            {
                const size_t eq = a.find('=');
                const std::string n = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
                if(eq != std::string::npos)
                    value = a.substr(eq + 1), hasValue = true;
                for(const Option& o : _options)
                    if(o.name == n)
                        opt = &o;
            }
            else if(a.size() == 2 && a[0] == '-')
            {
                for(const Option& o : _options)
                    if(o.shortName == a[1])
                        opt = &o;
            }
            if(!opt)
                return fail("unrecognised option '" + a + "'", error);
            if(!hasValue)
            {
                if(i + 1 >= argc)
                    return fail("the required argument for option '--" + opt->name + "' is missing", error);
                value = argv[++i];
            }
            try
            {
                opt->set(value);
                if(opt->multitoken)
                    while(i + 1 < argc && !(argv[i + 1][0] == '-' && (argv[i + 1][1] == '-' || std::isalpha((unsigned char)argv[i + 1][1]))))
                        opt->set(argv[++i]);
            }
            catch(const std::exception& e)
            {
                return fail(e.what(), error);
            }
            seen[opt->name] = true;
        }
        for(const Option& o : _options)
            if(o.required && !seen[o.name])
                return fail("the option '--" + o.name + "' is required but missing", error);
        return true;
    }
    void printParams(std::ostream& os) const
    {
        os << "Program called with the following parameters:" << std::endl;
        for(const Option& o : _options)
            if(!o.hidden)
                os << " * " << o.name << " = " << o.dflt << std::endl;
    }
    void refreshDefaults(const std::function<std::string(const std::string&)>& current)
    {
        for(Option& o : _options)
            o.dflt = current(o.name);
    }

  private:
    std::string _desc;
    std::vector<Option> _options;

    static void parse(const std::string&, const std::string& v, std::string* t) { *t = v; }
    static void parse(const std::string& n, const std::string& v, int* t)
    {
        char* end = nullptr;
        const long x = std::strtol(v.c_str(), &end, 10);
        if(end == v.c_str() || *end)
            throw std::runtime_error("the argument ('" + v + "') for option '--" + n + "' is invalid");
        *t = (int)x;
    }
    static void parse(const std::string& n, const std::string& v, float* t)
    {
        char* end = nullptr;
        *t = std::strtof(v.c_str(), &end);
        if(end == v.c_str() || *end)
            throw std::runtime_error("the argument ('" + v + "') for option '--" + n + "' is invalid");
    }
    static void parse(const std::string& n, const std::string& v, double* t)
    {
        char* end = nullptr;
        *t = std::strtod(v.c_str(), &end);
        if(end == v.c_str() || *end)
            throw std::runtime_error("the argument ('" + v + "') for option '--" + n + "' is invalid");
    }
    static void parse(const std::string& n, const std::string& v, bool* t)
    {
        std::string s = v;
        for(char& c : s)
            c = (char)std::tolower(c);
        if(s == "1" || s == "true" || s == "on" || s == "yes")
            *t = true;
        else if(s == "0" || s == "false" || s == "off" || s == "no")
            *t = false;
        else
            throw std::runtime_error("the argument ('" + v + "') for option '--" + n + "' is invalid. Valid choices are 'on|off', 'yes|no', '1|0' and 'true|false'");
    }
    bool fail(const std::string& msg, bool& error)
    {
        error = true;
        std::cerr << "ERROR: " << msg << std::endl << "Usage:\n\n";
        printHelp();
        return false;
    }
    void printHelp() const
    {
        std::cout << _desc << std::endl << std::endl;
        for(const Option& o : _options)
        {
            if(o.hidden)
                continue;
            std::cout << "  ";
            if(o.shortName)
                std::cout << "-" << o.shortName << " [ --" << o.name << " ]";
            else
                std::cout << "--" << o.name;
            std::cout << " arg";
            if(!o.required)
                std::cout << " (=" << o.dflt << ")";
            std::cout << "\n        " << o.help << std::endl;
        }
    }
};

} // namespace avdm_host
