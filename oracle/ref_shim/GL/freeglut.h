// ORACLE / TEST INFRASTRUCTURE ONLY — stand-in for <GL/freeglut.h>: just the type names
// the reference's graphic/graphictool.h mentions in its class declaration (the header is
// pulled in by monoslam.h; graphictool.cpp itself is NOT compiled).
#ifndef SL2_REF_SHIM_FREEGLUT
#define SL2_REF_SHIM_FREEGLUT
typedef unsigned int GLuint;
typedef struct GLUquadric GLUquadricObj;
#endif
