// mcd_inst.hip — one unit of kernel instantiations (see mcd_instances.hpp); compiled with -DMCD_INST_UNIT=1 .. MCD_INST_UNITS.
#include "mcd_launch.hpp"

#ifndef MCD_INST_UNIT
#error "compile with -DMCD_INST_UNIT=<n> (mocodad_amd/build.py does)"
#endif

#define MCD_UNIT_IS(n) (MCD_INST_UNIT == n)
#if MCD_UNIT_IS(1)
#define MCD_U1(...) __VA_ARGS__
#else
#define MCD_U1(...)
#endif
#if MCD_UNIT_IS(2)
#define MCD_U2(...) __VA_ARGS__
#else
#define MCD_U2(...)
#endif
#if MCD_UNIT_IS(3)
#define MCD_U3(...) __VA_ARGS__
#else
#define MCD_U3(...)
#endif
#if MCD_UNIT_IS(4)
#define MCD_U4(...) __VA_ARGS__
#else
#define MCD_U4(...)
#endif
#if MCD_UNIT_IS(5)
#define MCD_U5(...) __VA_ARGS__
#else
#define MCD_U5(...)
#endif
#if MCD_UNIT_IS(6)
#define MCD_U6(...) __VA_ARGS__
#else
#define MCD_U6(...)
#endif
#if MCD_UNIT_IS(7)
#define MCD_U7(...) __VA_ARGS__
#else
#define MCD_U7(...)
#endif
#if MCD_UNIT_IS(8)
#define MCD_U8(...) __VA_ARGS__
#else
#define MCD_U8(...)
#endif
#if MCD_UNIT_IS(9)
#define MCD_U9(...) __VA_ARGS__
#else
#define MCD_U9(...)
#endif
#if MCD_UNIT_IS(10)
#define MCD_U10(...) __VA_ARGS__
#else
#define MCD_U10(...)
#endif
#if MCD_UNIT_IS(11)
#define MCD_U11(...) __VA_ARGS__
#else
#define MCD_U11(...)
#endif
#if MCD_UNIT_IS(12)
#define MCD_U12(...) __VA_ARGS__
#else
#define MCD_U12(...)
#endif
#if MCD_UNIT_IS(13)
#define MCD_U13(...) __VA_ARGS__
#else
#define MCD_U13(...)
#endif
#if MCD_UNIT_IS(14)
#define MCD_U14(...) __VA_ARGS__
#else
#define MCD_U14(...)
#endif
#if MCD_UNIT_IS(15)
#define MCD_U15(...) __VA_ARGS__
#else
#define MCD_U15(...)
#endif
#if MCD_UNIT_IS(16)
#define MCD_U16(...) __VA_ARGS__
#else
#define MCD_U16(...)
#endif
#if MCD_UNIT_IS(17)
#define MCD_U17(...) __VA_ARGS__
#else
#define MCD_U17(...)
#endif
#if MCD_UNIT_IS(18)
#define MCD_U18(...) __VA_ARGS__
#else
#define MCD_U18(...)
#endif
#if MCD_UNIT_IS(19)
#define MCD_U19(...) __VA_ARGS__
#else
#define MCD_U19(...)
#endif
#if MCD_UNIT_IS(20)
#define MCD_U20(...) __VA_ARGS__
#else
#define MCD_U20(...)
#endif
#if MCD_UNIT_IS(21)
#define MCD_U21(...) __VA_ARGS__
#else
#define MCD_U21(...)
#endif
#if MCD_UNIT_IS(22)
#define MCD_U22(...) __VA_ARGS__
#else
#define MCD_U22(...)
#endif
#if MCD_UNIT_IS(23)
#define MCD_U23(...) __VA_ARGS__
#else
#define MCD_U23(...)
#endif
#if MCD_UNIT_IS(24)
#define MCD_U24(...) __VA_ARGS__
#else
#define MCD_U24(...)
#endif
#if MCD_UNIT_IS(25)
#define MCD_U25(...) __VA_ARGS__
#else
#define MCD_U25(...)
#endif
#if MCD_INST_UNITS != 25
#error "add the MCD_U<n> selectors of the new units"
#endif

namespace mcd {

#ifdef MCD_FAST_T       // developer builds (see mcd_api.hip): unit 1 holds the one trajectory kernel and its encoders, the rest is empty
#if MCD_UNIT_IS(1)
template int launch_score_t<MCD_FAST_T, MCD_FAST_NB, MCD_FAST_MINW, false>(ScoreParams&, hipStream_t, bool*);
template int launch_cond_fast_t<MCD_FAST_T, MCD_FAST_NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
template int launch_cond_unet_t<MCD_FAST_T, MCD_FAST_NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
#ifdef MCD_FAST_TILED
template int launch_score_tiled_t<MCD_FAST_TILED, tl_nb(MCD_FAST_TILED), false>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
#endif
#ifdef MCD_FAST_TILED_COND
template int launch_score_tiled_t<MCD_FAST_TILED_COND, tl_nb(MCD_FAST_TILED_COND), false, true>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
#endif
#endif
#else
#define MCD_DEF_SCORE(unit, T, NB, MINW, LT) MCD_U##unit(template int launch_score_t<T, NB, MINW, LT>(ScoreParams&, hipStream_t, bool*);)
#define MCD_DEF_COND_FAST(unit, T, NB) \
    MCD_U##unit(template int launch_cond_fast_t<T, NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);)
#define MCD_DEF_COND_UNET(unit, T, NB) \
    MCD_U##unit(template int launch_cond_unet_t<T, NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);)
#define MCD_DEF_TILED(unit, TP, NB, LT) \
    MCD_U##unit(template int launch_score_tiled_t<TP, NB, LT>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);)
MCD_SCORE_INSTANCES(MCD_DEF_SCORE)
MCD_COND_FAST_INSTANCES(MCD_DEF_COND_FAST)
MCD_COND_UNET_INSTANCES(MCD_DEF_COND_UNET)
MCD_TILED_INSTANCES(MCD_DEF_TILED)
#define MCD_DEF_TILED_COND(unit, TP, NB) \
    MCD_U##unit(template int launch_score_tiled_t<TP, NB, false, true>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);)
MCD_TILED_COND_INSTANCES(MCD_DEF_TILED_COND)
#endif

}  // namespace mcd
