// ParameterIO.cpp -- see ParameterIO.h (behaviour of the reference's ParameterIO.cpp:21-119).
#include "ParameterIO.h"
#include "Error.h"
#include <cstring>

void CParameterIO::PushParamPair(const char *name, const char *format, void *value_loc)
{
    m_entries.push_back(Entry{name, format, value_loc});
}

void CParameterIO::ReadParamPair(const char *name, const char *value, bool warn_bad_name)
{
    for (Entry &e : m_entries)
        if (e.name == name) {
            sscanf(value, e.format.c_str(), e.loc);
            return;
        }
    if (strcmp(name, "script") == 0 || strcmp(name, "cd") == 0)
        return;
    if (warn_bad_name)
        fprintf(stderr, "Warning: CParameterIO::ReadParamPair: unknown parameter %s\n", name);
}

void CParameterIO::ReadFromStream(FILE *stream)
{
    char name[1024], value[1024];
    while (fscanf(stream, "%1023s %1023s", name, value) == 2)
        ReadParamPair(name, value);
}

void CParameterIO::ReadFromFile(const char *filename)
{
    FILE *stream = fopen(filename, "r");
    if (!stream)
        throw CError("CParameterIO::ReadFromFile: could not open %s", filename);
    ReadFromStream(stream);
    fclose(stream);
}

void CParameterIO::ReadFromCommandLine(int argc, const char *argv[], bool warn_bad_name)
{
    for (int i = 0; i + 1 < argc; i += 2)
        ReadParamPair(argv[i], argv[i + 1], warn_bad_name);
}

void CParameterIO::WriteToStream(FILE *stream, bool single_line)
{
    for (const Entry &e : m_entries) {
        char kind = e.format.back();
        if (kind == 's') {
            const char *s = static_cast<const char *>(e.loc);
            if (!s[0])
                continue;
            fprintf(stream, "%s ", e.name.c_str());
            fprintf(stream, e.format.c_str(), s);
        } else if (kind == 'f') {
            fprintf(stream, "%s ", e.name.c_str());
            fprintf(stream, e.format.c_str(), *static_cast<const float *>(e.loc));
        } else if (kind == 'd' || kind == 'x') {
            fprintf(stream, "%s ", e.name.c_str());
            fprintf(stream, e.format.c_str(), *static_cast<const int *>(e.loc));
        } else {
            throw CError("CParamaterIO: illegal format '%s'\n", e.format.c_str());
        }
        fputs(single_line ? " " : "\n", stream);
    }
    if (single_line)
        fputs("\n", stream);
}

void CParameterIO::WriteToFile(const char *filename)
{
    bool append = filename[0] == '+';
    if (append)
        filename++;
    FILE *stream = fopen(filename, append ? "a" : "w");
    if (!stream)
        throw CError("CParameterIO::WriteToFile: could not open %s", filename);
    WriteToStream(stream);
    if (append)
        fputs("\n", stream);
    fclose(stream);
}
