// ParameterIO.h -- name / value parameter registry, the interface of the reference's
// CParameterIO (ParameterIO.h:32-57): every parameter is registered once with its name, a
// printf/scanf format and the address of the variable; values are read from argv-style pairs
// or from a file of "name value" pairs and written back one per line.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

class CParameterIO {
public:
    void PushParamPair(const char *name, const char *format, void *value_loc);
    void ReadFromStream(FILE *stream);
    void ReadFromFile(const char *filename);
    void ReadFromCommandLine(int argc, const char *argv[], bool warn_bad_name = true);
    void WriteToStream(FILE *stream, bool single_line = false);
    void WriteToFile(const char *filename);   // a leading '+' appends (ParameterIO.cpp:81-85)

protected:
    void ReadParamPair(const char *name, const char *value, bool warn_bad_name = true);
    struct Entry { std::string name, format; void *loc; };
    std::vector<Entry> m_entries;
};

#define PPPS(x) PushParamPair(#x, "%s", &x)
#define PPPF(x) PushParamPair(#x, "%f", &x)
#define PPPD(x) PushParamPair(#x, "%d", &x)
#define PPPH(x) PushParamPair(#x, "0x%08x", &x)
