// ParameterIO.h -- the name / value registry behind the script grammar and the result files.
// Interface names follow the reference's CParameterIO (ParameterIO.h:32-57: PushParamPair, ReadFrom*,
// WriteTo*, the PPP* macros); the implementation is a typed table: every parameter is registered once
// with the address of a std::string, float or int, and knows how to parse and print itself.  What the
// result files look like is fixed by byte parity with the reference's (one "name value" line per
// parameter, floats as %f, flags as 0x%08x, empty strings skipped, ParameterIO.cpp:87-119).
#pragma once
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

class CParameterIO {
public:
    void PushParamPair(const char *name, std::string *text);
    void PushParamPair(const char *name, float *real);
    void PushParamPair(const char *name, int *integer, bool hex = false);

    void ReadFromStream(FILE *stream);
    void ReadFromFile(const char *filename);
    void ReadFromCommandLine(int argc, const char *argv[], bool warn_bad_name = true);
    void WriteToStream(FILE *stream, bool single_line = false);
    void WriteToFile(const char *filename);   // "+name" appends to name

private:
    enum class Kind { Text, Real, Integer, Hex };
    struct Param {
        std::string name;
        Kind kind;
        void *loc;
        void Parse(const std::string &value) const;
        bool Print(std::string &out) const;    // false: nothing to print (empty text)
    };
    void Add(const char *name, Kind kind, void *loc);
    void Assign(const std::string &name, const std::string &value, bool warn_bad_name);

    std::vector<Param> m_params;                              // registration order = output order
    std::unordered_map<std::string, size_t> m_index;          // first registration of a name wins
};

#define PPPS(x) PushParamPair(#x, &x)
#define PPPF(x) PushParamPair(#x, &x)
#define PPPD(x) PushParamPair(#x, &x)
#define PPPH(x) PushParamPair(#x, &x, true)
