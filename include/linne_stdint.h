/* Fixed-width integer types used by the LINNE API (boundary file; replaces the reference's
 * include/linne_stdint.h:1-11, which is a plain <stdint.h> wrapper). */
#ifndef LINNE_STDINT_H_INCLUDED
#define LINNE_STDINT_H_INCLUDED
#include <stdint.h>
#endif
