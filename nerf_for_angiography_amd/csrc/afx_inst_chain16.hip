// afx_inst_chain16.hip — explicit instantiations of the 16-bit chain kernels for ONE width and direction
// (-DAFX_INST_F=64|128|256 -DAFX_INST_BWD=0|1|2; 2 = the split phases and the tanh / sine forward kernels): the library is built from several translation units in parallel.
#define AFX_TEMPLATES_ONLY
#include "afx_kernels_f32.hip"
#include "afx_kernels_bf16.hip"
#include "afx_inst.h"
#if AFX_INST_BWD == 2
AFX_CHAIN16_PHASES(AFX_CHAIN16_PH_DEF, AFX_INST_F)
AFX_CHAIN16_ACTS(AFX_CHAIN16_ACT_DEF, AFX_INST_F)
#elif AFX_INST_BWD
AFX_CHAIN16_BWD(AFX_CHAIN16_DEF, AFX_INST_F)
#else
AFX_CHAIN16_FWD(AFX_CHAIN16_DEF, AFX_INST_F)
#endif
