// ORACLE / TEST INFRASTRUCTURE ONLY — stand-in for <boost/thread.hpp>: the reference's
// framegrabber/framegrabber.h (pulled in by monoslam.h) declares one boost::mutex member;
// the grabbers themselves are NOT compiled.
#ifndef SL2_REF_SHIM_BOOST_THREAD
#define SL2_REF_SHIM_BOOST_THREAD
#include <mutex>
namespace boost {
typedef std::mutex mutex;
}
#endif
