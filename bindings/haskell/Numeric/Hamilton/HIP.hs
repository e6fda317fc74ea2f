{-# LANGUAGE DataKinds #-}
{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE RankNTypes #-}
{-# LANGUAGE ScopedTypeVariables #-}
{-# LANGUAGE TypeApplications #-}

-- |
-- Module      : Numeric.Hamilton.HIP
-- Description : FFI shim from Numeric.Hamilton onto libhamk.so (MI355X / gfx950)
--
-- SOURCE DELIVERABLE, NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no
-- GHC/cabal (SURVEY.md F3).  It is the binding a maintainer of mstksg/hamilton adds
-- next to src/Numeric/Hamilton.hs; see INTEGRATION.md for where it hooks in.
--
-- What it does:
--
--   * 'Traced' is a 'RealFloat' instance whose arithmetic records an expression
--     tape.  'traceSystem' instantiates the user's rank-2 polymorphic functions
--     (the arguments of 'mkSystem' / 'mkSystem'', Hamilton.hs:201-254) at 'Traced'
--     once and ships the tapes to @hamk_system_create@.
--   * the batched entry points ('hamEqsBatch', 'stepHamBatch', 'rk4StepsBatch',
--     'evolveHamEnsemble', ...) run ensembles of phases on the GPU; an ensemble is a
--     pair of storable vectors in structure-of-arrays order (q[j*B + i]).
--   * comparisons on 'Traced' ('Ord', 'RealFrac', 'isNaN', ...) throw
--     'UntraceableFunction': such systems keep the pure ad+hmatrix closures.
--   * what crosses the ABI is the CANONICAL form of the recording ('canonical': only the
--     values the outputs depend on, numbered depth-first post-order from the outputs) with
--     the folding rules of the other host shims (hamilton_amd/tracer.py, include/hamilton.hpp):
--     a function of the expression alone -- independent of the order in which lazy
--     evaluation happened to force the recording -- and therefore byte-identical to what
--     the Python and C++ shims send for the same system (tests/test_recorders.py there).
module Numeric.Hamilton.HIP
  ( HipSystem
  , traceSystem
  , traceSystem'
    -- * the library's choices as an explicit record (hamk.h @hamk_options@; 0 = HAMK_AUTO)
  , Options (..)
  , defaultOptions
  , traceSystemWith
  , traceSystemWith'
  , Ensemble (..)
  , toPhaseBatch
  , fromPhaseBatch
  , hamEqsBatch
  , hamiltonianBatch
  , stepHamBatch
  , iterateStepHamBatch
  , rk4StepsBatch
  , evolveHamEnsemble
    -- * ensembles resident in HBM, several GPUs from one process
  , DeviceEnsemble
  , setDevice
  , uploadEnsemble
  , sampleEnsembleDevice
  , setEnsembleSize
  , downloadEnsemble
  , rk4StepsDevice
  , stepHamDevice
  , iterateStepHamDevice
  , synchronize
  , gatherEnsembles
    -- * one process per GPU: the final all-gather over RCCL (hamk_comm_*)
  , Comm
  , CommId
  , commUniqueId
  , commCreate
  , commDestroy
  , allGatherEnsemble
    -- * which GSL binding of hmatrix-gsl 'stepHam' / 'evolveHam' reproduce (hamk.h)
  , GslApi (..)
  , setGslApi
    -- * fixed-step launches that check their own energy invariant; ensemble checkpoints
  , rk4StepsCheckedDevice
  , statusDrift
  , saveCheckpoint
  , loadCheckpoint
  , CheckpointInfo (..)
  , checkpointInfo
  , UntraceableFunction (..)
  ) where

import Control.Exception
import Control.Monad
import Data.IORef
import Data.Int
import Data.Word
import qualified Data.Map.Strict as M
import qualified Data.Sequence as Seq
import Data.Foldable (toList)
import GHC.Float (castDoubleToWord64)
import qualified Data.Vector.Sized as V
import qualified Data.Vector.Storable as VS
import qualified Data.Vector.Storable.Mutable as VSM
import Foreign
import Foreign.C.String
import Foreign.C.Types
import GHC.TypeLits
import Data.Proxy
import System.IO.Unsafe (unsafePerformIO)

-- ---------------------------------------------------------------------------
-- C ABI (include/hamk.h) -- one declaration per entry point used here
-- ---------------------------------------------------------------------------
data HamkSystem
data HamkComm

-- struct hamk_op { int32 op, a, b, _pad; double c; }  (24 bytes)
data Op = Op !Int32 !Int32 !Int32 !Double

instance Storable Op where
  sizeOf _ = 24
  alignment _ = 8
  peek p = Op <$> peekByteOff p 0 <*> peekByteOff p 4 <*> peekByteOff p 8 <*> peekByteOff p 16
  poke p (Op o a b c) = pokeByteOff p 0 o >> pokeByteOff p 4 a >> pokeByteOff p 8 b
                     >> pokeByteOff p 12 (0 :: Int32) >> pokeByteOff p 16 c

foreign import ccall safe "hamk_system_create_ex"
  c_system_create_ex :: Int32 -> Int32 -> Ptr Double -> Ptr Op -> Int32 -> Ptr Int32
                     -> Ptr Op -> Int32 -> Int32 -> Int32 -> Ptr Options -> Ptr (Ptr HamkSystem) -> IO CInt

-- | @hamk_options@ (include/hamk.h): everything the library otherwise decides for itself when it specialises its
--   kernels for a system.  0 (@HAMK_AUTO@) leaves a choice to the library; the numeric values are the header's
--   (@HAMK_MAP_*@, @HAMK_AD_*@, @HAMK_BODY_*@, @HAMK_TRIG_*@, @HAMK_ON@ = 1 / @HAMK_OFF@ = 2).
data Options = Options
  { optMapping, optAdMode, optRk4Body, optRkfBody, optTrig, optGslApi, optSelfCheck, optBuild
  , optRk4MinWaves, optKReassoc, optRk4Park, optMaxSubsteps, optCache :: Int32
  , optLanesPerTrajectory :: Int32   -- ^ OUTPUT of @hamk_system_get_options@: 1, 4, 16, 32 or 64
  , optRkfPark :: Int32              -- ^ QUAD mapping: the adaptive stepper's vectors parked in LDS / a private array (ON / OFF / AUTO);
                                     --   lane mapping: follows 'optRkfBody' (reported; a contradicting value is refused)
  , optEnsembleSize :: Int64         -- ^ size of the WHOLE ensemble the handle's launches are pieces of: the mapping is chosen
                                     --   for it, so any shard layout reproduces the one-launch bits; 0 = per launch
  }

defaultOptions :: Options
defaultOptions = Options 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0

-- | @HAMK_OPTIONS_VERSION@ of include/hamk.h: the layout revision this instance writes (round 5: @wave_blocked@ removed).
optionsVersion :: Word32
optionsVersion = 0x484b0005

instance Storable Options where
  sizeOf _ = 128                                   -- uint32 size, uint32 version, 13 choices, lanes_per_trajectory, rkf_park, pad, int64 ensemble_size, reserved[12]
  alignment _ = 8
  peek p = Options <$> f 8 <*> f 12 <*> f 16 <*> f 20 <*> f 24 <*> f 28 <*> f 32 <*> f 36 <*> f 40 <*> f 44
                   <*> f 48 <*> f 52 <*> f 56 <*> f 60 <*> f 64 <*> peekByteOff p 72
    where f = peekByteOff p
  poke p (Options a b c d e g h i k l m' n' o' lanes park ens) = do
    mapM_ (\off -> pokeByteOff p off (0 :: Int32)) [0, 4 .. 124]
    pokeByteOff p 0 (128 :: Word32)
    pokeByteOff p 4 optionsVersion
    mapM_ (\(off, v) -> pokeByteOff p off v) (zip [8, 12 ..] [a, b, c, d, e, g, h, i, k, l, m', n', o', lanes, park])
    pokeByteOff p 72 ens
foreign import ccall "&hamk_system_destroy"
  p_system_destroy :: FunPtr (Ptr HamkSystem -> IO ())
foreign import ccall safe "hamk_to_phase_batch"
  c_to_phase :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Int32 -> IO CInt
foreign import ccall safe "hamk_from_phase_batch"
  c_from_phase :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_observe_batch"
  c_observe :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double
            -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_hameqs_batch"
  c_hameqs :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Int32
           -> Int32 -> IO CInt
foreign import ccall safe "hamk_rk4_steps"
  c_rk4_steps :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Double -> Int32 -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_step_ham_batch"
  c_step_ham :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Double -> Ptr Int32 -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_step_ham_iterate"
  c_step_ham_iterate :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Double -> Int32 -> Int32 -> Ptr Double -> Ptr Double
                     -> Ptr Int32 -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_evolve_ham_batch"
  c_evolve_ham :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Int32 -> Ptr Double -> Ptr Double -> Ptr Double
               -> Double -> Double -> Double -> Ptr Int32 -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall unsafe "hamk_last_error"
  c_last_error :: IO CString
foreign import ccall safe "hamk_synchronize"
  c_synchronize :: Ptr HamkSystem -> IO CInt
foreign import ccall unsafe "hamk_set_device"
  c_set_device :: Int32 -> IO CInt
foreign import ccall unsafe "hamk_device_malloc"
  c_device_malloc :: Ptr (Ptr Double) -> Int64 -> IO CInt
foreign import ccall "&hamk_device_free"
  p_device_free :: FunPtr (Ptr Double -> IO ())
foreign import ccall safe "hamk_memcpy"
  c_memcpy :: Ptr Double -> Ptr Double -> Int64 -> Int32 -> IO CInt
foreign import ccall safe "hamk_gather_batch"
  c_gather :: Int32 -> Int32 -> Ptr Int64 -> Ptr (Ptr Double) -> Ptr Double -> Int32 -> IO CInt
foreign import ccall unsafe "hamk_system_set_ensemble_size"
  c_system_set_ensemble_size :: Ptr HamkSystem -> Int64 -> IO CInt
foreign import ccall safe "hamk_comm_unique_id"
  c_comm_unique_id :: Ptr Word8 -> IO CInt
foreign import ccall safe "hamk_comm_create"
  c_comm_create :: Ptr Word8 -> Int32 -> Int32 -> Ptr (Ptr HamkComm) -> IO CInt
foreign import ccall safe "hamk_comm_allgather_batch"
  c_comm_allgather :: Ptr HamkComm -> Int32 -> Ptr Int64 -> Ptr Double -> Ptr Double -> IO CInt
foreign import ccall safe "hamk_comm_destroy"
  c_comm_destroy :: Ptr HamkComm -> IO CInt
foreign import ccall safe "hamk_sample_batch"
  c_sample_batch :: Ptr HamkSystem -> Int64 -> Int64 -> Word64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double
                 -> Ptr Double -> Ptr Double -> Int32 -> IO CInt
foreign import ccall unsafe "hamk_system_set_gsl_api"
  c_set_gsl_api :: Ptr HamkSystem -> Int32 -> IO CInt
foreign import ccall safe "hamk_rk4_steps_checked"
  c_rk4_steps_checked :: Ptr HamkSystem -> Int64 -> Ptr Double -> Ptr Double -> Double -> Int32 -> Double -> Ptr Int32 -> Int32 -> IO CInt
foreign import ccall safe "hamk_checkpoint_write"
  c_ck_write :: CString -> Int32 -> Int64 -> Ptr Double -> Ptr Double -> Int32 -> Int64 -> Word64 -> Double -> IO CInt
foreign import ccall safe "hamk_checkpoint_info"
  c_ck_info :: CString -> Ptr Int32 -> Ptr Int64 -> Ptr Int64 -> Ptr Word64 -> Ptr Double -> IO CInt
foreign import ccall safe "hamk_checkpoint_read"
  c_ck_read :: CString -> Int32 -> Int64 -> Ptr Double -> Ptr Double -> Int32 -> IO CInt

memHost, memDevice :: Int32
memHost = 0
memDevice = 1

check :: String -> CInt -> IO ()
check what rc = when (rc /= 0) $ do
  msg <- c_last_error >>= peekCString
  -- the reference raises from pure code here (Hamilton.hs:425,444,462; hmatrix/GSL exceptions)
  throwIO . ErrorCall $ what ++ ": libhamk error " ++ show rc ++ ": " ++ msg

-- ---------------------------------------------------------------------------
-- recording number type
-- ---------------------------------------------------------------------------
data UntraceableFunction = UntraceableFunction String deriving Show
instance Exception UntraceableFunction

-- | A traced value: either a late-bound constant or the id of a tape value.
data Traced = K !Double | T !(IORef TapeSt) !Int32

-- the recording so far: ops in emission order + the hash-consing table (identical
-- subexpressions share one value)
data TapeSt = TapeSt !(Seq.Seq Op) !(M.Map (Int32, Int32, Int32, Word64) Int32)

opConst, opInput, opAdd, opSub, opMul, opDiv, opNeg, opRecip, opSin, opCos, opTan, opAsin, opAcos,
  opAtan, opSinh, opCosh, opTanh, opExp, opLog, opSqrt, opPowC, opPowI, opPow, opAtan2, opAsinh,
  opAcosh, opAtanh, opAbs, opSignum :: Int32
[ opConst, opInput, opAdd, opSub, opMul, opDiv, opNeg, opRecip, opSin, opCos, opTan, opAsin, opAcos
  , opAtan, opSinh, opCosh, opTanh, opExp, opLog, opSqrt, opPowC, opPowI, opPow, opAtan2, opAsinh
  , opAcosh, opAtanh, opAbs, opSignum ] = [0 .. 28]

emit :: IORef TapeSt -> Op -> Int32
emit ref o@(Op c a b d) = unsafePerformIO $ atomicModifyIORef' ref $ \st@(TapeSt ops memo) ->
  let key = (c, a, b, castDoubleToWord64 d)
  in case M.lookup key memo of
       Just i -> (st, i)
       Nothing -> let i = fromIntegral (Seq.length ops) in (TapeSt (ops Seq.|> o) (M.insert key i memo), i)
{-# NOINLINE emit #-}

opAt :: IORef TapeSt -> Int32 -> Op
opAt ref i = unsafePerformIO $ do TapeSt ops _ <- readIORef ref; return (Seq.index ops (fromIntegral i))
{-# NOINLINE opAt #-}

onTape :: IORef TapeSt -> Traced -> Int32
onTape ref (K c) = emit ref (Op opConst 0 0 c)
onTape _ (T _ i) = i

tapeOf :: Traced -> Traced -> IORef TapeSt
tapeOf (T r _) _ = r
tapeOf _ (T r _) = r
tapeOf _ _ = error "Numeric.Hamilton.HIP: two constants have no tape"

-- -c folds, -(-x) = x
neg :: Traced -> Traced
neg (K c) = K (negate c)
neg (T r i) = case opAt r i of
  Op o a _ _ | o == opNeg -> T r a
  _ -> T r (emit r (Op opNeg i 0 0))

-- constants fold; the exact identities (x+0, 0+x, x-0, 0-x, x*1, 1*x, x*(-1), (-1)*x, x/1, 1/x) are
-- applied -- never anything that could change a result bit; operands stay in the order written
bin :: Int32 -> (Double -> Double -> Double) -> Traced -> Traced -> Traced
bin _ f (K a) (K b) = K (f a b)
bin o _ a b
  | o == opAdd, K 0 <- a = b
  | o == opAdd, K 0 <- b = a
  | o == opSub, K 0 <- b = a
  | o == opSub, K 0 <- a = neg b
  | o == opMul, K 1 <- a = b
  | o == opMul, K 1 <- b = a
  | o == opMul, K (-1) <- a = neg b
  | o == opMul, K (-1) <- b = neg a
  | o == opDiv, K 1 <- b = a
  | o == opDiv, K 1 <- a = let r = tapeOf a b in T r (emit r (Op opRecip (onTape r b) 0 0))
  | otherwise = let r = tapeOf a b
                    ia = onTape r a          -- first operand first: same emission order as the other shims
                    ib = ia `seq` onTape r b
                in T r (emit r (Op o ia ib 0))

un :: Int32 -> (Double -> Double) -> Traced -> Traced
un _ f (K a) = K (f a)
un o _ (T r i) = T r (emit r (Op o i 0 0))

-- x ^ k of the other shims (`powi`): k = 0 folds to 1, k = 1 to x
powI :: Traced -> Int -> Traced
powI (K c) k = K (c ^^ k)
powI _ 0 = K 1
powI x 1 = x
powI (T r i) k = T r (emit r (Op opPowI i (fromIntegral k) 0))

untraceable :: String -> a
untraceable = throw . UntraceableFunction

instance Num Traced where
  (+) = bin opAdd (+)
  (-) = bin opSub (-)
  (*) = bin opMul (*)
  negate = neg
  abs = un opAbs abs             -- recorded: derivative signum x, as `ad` differentiates it
  signum = un opSignum signum
  fromInteger = K . fromInteger

instance Fractional Traced where
  (/) = bin opDiv (/)
  recip = un opRecip recip
  fromRational = K . fromRational

instance Floating Traced where
  pi = K pi
  exp = un opExp exp; log = un opLog log; sqrt = un opSqrt sqrt
  sin = un opSin sin; cos = un opCos cos; tan = un opTan tan
  asin = un opAsin asin; acos = un opAcos acos; atan = un opAtan atan
  sinh = un opSinh sinh; cosh = un opCosh cosh; tanh = un opTanh tanh
  asinh = un opAsinh asinh; acosh = un opAcosh acosh; atanh = un opAtanh atanh
  K a ** K b = K (a ** b)
  x@(T r i) ** K c
    | c == fromIntegral (round c :: Int) && abs c <= 64 = powI x (round c)     -- x ** 2 with x < 0 (Examples.hs:154)
    | otherwise = T r (emit r (Op opPowC i 0 c))
  a ** b = bin opPow (**) a b

instance Eq Traced where _ == _ = untraceable "(==)"
instance Ord Traced where compare _ _ = untraceable "compare"
instance Real Traced where toRational _ = untraceable "toRational"
instance RealFrac Traced where properFraction _ = untraceable "properFraction"
instance RealFloat Traced where
  floatRadix _ = 2; floatDigits _ = 53; floatRange _ = (-1021, 1024)
  decodeFloat _ = untraceable "decodeFloat"; encodeFloat m e = K (encodeFloat m e)
  isNaN _ = untraceable "isNaN"; isInfinite _ = untraceable "isInfinite"
  isDenormalized _ = untraceable "isDenormalized"; isNegativeZero _ = untraceable "isNegativeZero"
  isIEEE _ = True
  atan2 = bin opAtan2 atan2

-- ---------------------------------------------------------------------------
-- System construction
-- ---------------------------------------------------------------------------
-- | Opaque handle next to (not instead of) the reference's 'System' record.
data HipSystem (m :: Nat) (n :: Nat) = HipSystem !(ForeignPtr HamkSystem)

record :: Int -> ([Traced] -> [Traced]) -> IO ([Op], [Int32])
record nIn fn = do
  ref <- newIORef (TapeSt Seq.empty M.empty)
  let ins = [T ref (emit ref (Op opInput (fromIntegral j) 0 0)) | j <- [0 .. nIn - 1]]
  mapM_ (\(T _ i) -> evaluate i) ins
  outs <- mapM (evaluate . onTape ref) (fn ins)
  TapeSt ops _ <- readIORef ref
  return (canonical (toList ops) outs)

-- | Canonical form of a recording: only the values the outputs depend on, numbered depth-first
--   post-order from the outputs (first operand before second, outputs in order).  The same
--   function as @Tape.canonical@ (hamilton_amd/tracer.py) and @Tape::canonical@ (include/hamilton.hpp).
canonical :: [Op] -> [Int32] -> ([Op], [Int32])
canonical ops outs = (reverse rev, map (newId M.!) outs)
  where
    arr = Seq.fromList ops
    kids (Op o a b _)
      | o == opConst || o == opInput = []
      | o `elem` [opAdd, opSub, opMul, opDiv, opPow, opAtan2] = [a, b]
      | otherwise = [a]
    (newId, rev, _) = foldl visit (M.empty, [], 0 :: Int32) outs
    visit st@(seen, acc, next) node
      | M.member node seen = st
      | otherwise =
          let o@(Op c a b d) = Seq.index arr (fromIntegral node)
              ks = kids o
              (seen', acc', next') = foldl visit (seen, acc, next) ks
              a' = if null ks then a else seen' M.! a
              b' = if length ks == 2 then seen' M.! b else b
          in (M.insert node next' seen', Op c a' b' d : acc', next' + 1)

create :: forall m n. (KnownNat m, KnownNat n)
       => Options -> Int32 -> [Double] -> ([Traced] -> [Traced]) -> ([Traced] -> Traced) -> IO (HipSystem m n)
create opts uSpace inertia f u = do
  let m = fromIntegral (natVal (Proxy @m)); n = fromIntegral (natVal (Proxy @n))
  (fOps, fOuts) <- record n f
  (uOps, [uOut]) <- record (if uSpace == 1 then m else n) (pure . u)
  withArrayLen inertia $ \_ pI -> withArrayLen fOps $ \nf pF -> withArray fOuts $ \pFO ->
    withArrayLen uOps $ \nu pU -> alloca $ \pH -> with opts $ \pO -> do
      c_system_create_ex (fromIntegral m) (fromIntegral n) pI pF (fromIntegral nf) pFO
                         pU (fromIntegral nu) uOut uSpace pO pH >>= check "mkSystem"
      HipSystem <$> (peek pH >>= newForeignPtr p_system_destroy)

-- | 'mkSystem' (Hamilton.hs:201-225): potential over generalized coordinates.
traceSystem :: forall m n. (KnownNat m, KnownNat n)
            => V.Vector m Double
            -> (forall a. RealFloat a => V.Vector n a -> V.Vector m a)
            -> (forall a. RealFloat a => V.Vector n a -> a)
            -> IO (HipSystem m n)
traceSystem = traceSystemWith defaultOptions

-- | 'traceSystem' with the library's choices fixed by the caller.
traceSystemWith :: forall m n. (KnownNat m, KnownNat n)
                => Options
                -> V.Vector m Double
                -> (forall a. RealFloat a => V.Vector n a -> V.Vector m a)
                -> (forall a. RealFloat a => V.Vector n a -> a)
                -> IO (HipSystem m n)
traceSystemWith o w f u = create o 0 (V.toList w) (V.toList . f . unsafeSized) (u . unsafeSized)

-- | 'mkSystem'' (Hamilton.hs:238-254): potential over the underlying cartesian coordinates.
traceSystem' :: forall m n. (KnownNat m, KnownNat n)
             => V.Vector m Double
             -> (forall a. RealFloat a => V.Vector n a -> V.Vector m a)
             -> (forall a. RealFloat a => V.Vector m a -> a)
             -> IO (HipSystem m n)
traceSystem' = traceSystemWith' defaultOptions

-- | "traceSystem'" with the library's choices fixed by the caller.
traceSystemWith' :: forall m n. (KnownNat m, KnownNat n)
                 => Options
                 -> V.Vector m Double
                 -> (forall a. RealFloat a => V.Vector n a -> V.Vector m a)
                 -> (forall a. RealFloat a => V.Vector m a -> a)
                 -> IO (HipSystem m n)
traceSystemWith' o w f u = create o 1 (V.toList w) (V.toList . f . unsafeSized) (u . unsafeSized)

unsafeSized :: forall k a. KnownNat k => [a] -> V.Vector k a
unsafeSized xs = case V.fromList xs of
  Just v -> v
  Nothing -> error "Numeric.Hamilton.HIP: internal size mismatch"

-- ---------------------------------------------------------------------------
-- ensembles
-- ---------------------------------------------------------------------------
-- | B phases (or configs) of an n-coordinate system, structure of arrays:
--   element (j*B + i) is coordinate j of trajectory i.
data Ensemble (n :: Nat) = Ensemble
  { ensSize :: !Int
  , ensPositions :: !(VS.Vector Double)
  , ensMomenta :: !(VS.Vector Double)   -- momenta of a phase / velocities of a config
  }

withEns :: Ensemble n -> (Int64 -> Ptr Double -> Ptr Double -> IO a) -> IO a
withEns (Ensemble b q p) k = VS.unsafeWith q $ \pq -> VS.unsafeWith p $ \pp -> k (fromIntegral b) pq pp

newOut :: Int -> IO (VSM.IOVector Double)
newOut = VSM.new

-- | 'toPhase' (Hamilton.hs:279-284) on an ensemble of configs.
toPhaseBatch :: forall m n. KnownNat n => HipSystem m n -> Ensemble n -> IO (Ensemble n)
toPhaseBatch (HipSystem h) e = withForeignPtr h $ \s -> withEns e $ \b pq pv -> do
  out <- newOut (VS.length (ensPositions e))
  VSM.unsafeWith out $ \po -> c_to_phase s b pq pv po memHost >>= check "toPhase"
  Ensemble (ensSize e) (ensPositions e) <$> VS.unsafeFreeze out

-- | 'fromPhase' (Hamilton.hs:332-337); also returns the per-trajectory status words.
fromPhaseBatch :: forall m n. KnownNat n => HipSystem m n -> Ensemble n -> IO (Ensemble n, VS.Vector Int32)
fromPhaseBatch (HipSystem h) e = withForeignPtr h $ \s -> withEns e $ \b pq pp -> do
  out <- newOut (VS.length (ensPositions e)); st <- VSM.new (ensSize e)
  VSM.unsafeWith out $ \po -> VSM.unsafeWith st $ \ps -> c_from_phase s b pq pp po ps memHost >>= check "fromPhase"
  (,) <$> (Ensemble (ensSize e) (ensPositions e) <$> VS.unsafeFreeze out) <*> VS.unsafeFreeze st

-- | 'hamEqs' (Hamilton.hs:370-387): (dH/dp, -dH/dq) for every member.
hamEqsBatch :: forall m n. KnownNat n => HipSystem m n -> Ensemble n -> IO (VS.Vector Double, VS.Vector Double)
hamEqsBatch (HipSystem h) e = withForeignPtr h $ \s -> withEns e $ \b pq pp -> do
  dq <- newOut (VS.length (ensPositions e)); dp <- newOut (VS.length (ensPositions e))
  VSM.unsafeWith dq $ \a -> VSM.unsafeWith dp $ \c -> c_hameqs s b pq pp a c nullPtr memHost >>= check "hamEqs"
  (,) <$> VS.unsafeFreeze dq <*> VS.unsafeFreeze dp

-- | 'hamiltonian' (Hamilton.hs:353-361).
hamiltonianBatch :: forall m n. KnownNat n => HipSystem m n -> Ensemble n -> IO (VS.Vector Double)
hamiltonianBatch (HipSystem h) e = withForeignPtr h $ \s -> withEns e $ \b pq pp -> do
  out <- newOut (ensSize e)
  VSM.unsafeWith out $ \po -> c_observe s b pq pp nullPtr nullPtr po nullPtr memHost >>= check "hamiltonian"
  VS.unsafeFreeze out

inPlace :: Ensemble n -> (Int64 -> Ptr Double -> Ptr Double -> IO CInt) -> String -> IO (Ensemble n)
inPlace e k what = do
  q <- VS.thaw (ensPositions e); p <- VS.thaw (ensMomenta e)
  VSM.unsafeWith q $ \pq -> VSM.unsafeWith p $ \pp -> k (fromIntegral (ensSize e)) pq pp >>= check what
  Ensemble (ensSize e) <$> VS.unsafeFreeze q <*> VS.unsafeFreeze p

-- | 'stepHam' (Hamilton.hs:390-402) for every member: adaptive RKF45, GSL semantics.
stepHamBatch :: forall m n. KnownNat n => Double -> HipSystem m n -> Ensemble n -> IO (Ensemble n)
stepHamBatch r (HipSystem h) e = withForeignPtr h $ \s ->
  inPlace e (\b pq pp -> c_step_ham s b pq pp r nullPtr nullPtr memHost) "stepHam"

-- | @iterate (stepHam r s)@ (README.md:150; the demo's frame loop, app/Examples.hs:429): @k@ consecutive
--   'stepHam' calls in ONE launch, bit-identical to @k@ separate calls.
iterateStepHamBatch :: forall m n. KnownNat n => Double -> Int -> HipSystem m n -> Ensemble n -> IO (Ensemble n)
iterateStepHamBatch r k (HipSystem h) e = withForeignPtr h $ \s ->
  inPlace e (\b pq pp -> c_step_ham_iterate s b pq pp r (fromIntegral k) 0 nullPtr nullPtr nullPtr nullPtr memHost) "iterateStepHam"

-- | Classic fixed-step RK4 (no counterpart in the reference; SURVEY.md F1).
rk4StepsBatch :: forall m n. KnownNat n => Double -> Int -> HipSystem m n -> Ensemble n -> IO (Ensemble n)
rk4StepsBatch dt k (HipSystem h) e = withForeignPtr h $ \s ->
  inPlace e (\b pq pp -> c_rk4_steps s b pq pp dt (fromIntegral k) nullPtr memHost) "rk4Steps"

-- | 'evolveHam' (Hamilton.hs:433-462) for every member: one ensemble per requested time,
--   element 0 being the initial ensemble.
evolveHamEnsemble :: forall m n. KnownNat n => HipSystem m n -> Ensemble n -> [Double] -> IO [Ensemble n]
evolveHamEnsemble (HipSystem h) e ts
  | length ts < 2 = throwIO (ErrorCall "evolveHam: at least two solution times required (2 <= s)")
  | otherwise = withForeignPtr h $ \s -> withEns e $ \b pq pp -> withArrayLen ts $ \nt pts -> do
      let cnt = VS.length (ensPositions e)
      qo <- newOut (cnt * nt); po <- newOut (cnt * nt)
      VSM.unsafeWith qo $ \a -> VSM.unsafeWith po $ \c ->
        c_evolve_ham s b pq pp (fromIntegral nt) pts a c 0 0 0 nullPtr nullPtr memHost >>= check "evolveHam"
      qf <- VS.unsafeFreeze qo; pf <- VS.unsafeFreeze po
      return [Ensemble (ensSize e) (VS.slice (r * cnt) cnt qf) (VS.slice (r * cnt) cnt pf) | r <- [0 .. nt - 1]]

-- ---------------------------------------------------------------------------
-- ensembles resident in HBM (HAMK_MEM_DEVICE) and node-level sharding
-- ---------------------------------------------------------------------------
-- | An ensemble whose arrays live in device memory of the device that was current
--   ('setDevice') when it was uploaded; freed by the garbage collector through
--   @hamk_device_free@.  Steppers advance it in place and return at once: the launch is
--   asynchronous on the system's stream ('synchronize' waits).
data DeviceEnsemble (n :: Nat) = DeviceEnsemble
  { devSize :: !Int
  , devPositions :: !(ForeignPtr Double)
  , devMomenta :: !(ForeignPtr Double)
  }

-- | Select the GPU the calling thread's next calls run on (one 'HipSystem' per device).
setDevice :: Int -> IO ()
setDevice d = c_set_device (fromIntegral d) >>= check "setDevice"

deviceArray :: Int -> IO (ForeignPtr Double)
deviceArray count = alloca $ \pp -> do
  c_device_malloc pp (fromIntegral (8 * count)) >>= check "deviceMalloc"
  peek pp >>= newForeignPtr p_device_free

uploadEnsemble :: forall n. KnownNat n => Ensemble n -> IO (DeviceEnsemble n)
uploadEnsemble e = do
  let cnt = VS.length (ensPositions e)
  dq <- deviceArray cnt; dp <- deviceArray cnt
  withEns e $ \_ pq pp -> withForeignPtr dq $ \a -> withForeignPtr dp $ \b -> do
    c_memcpy a pq (fromIntegral (8 * cnt)) 0 >>= check "upload"     -- HAMK_COPY_H2D
    c_memcpy b pp (fromIntegral (8 * cnt)) 0 >>= check "upload"
  return (DeviceEnsemble (ensSize e) dq dp)

downloadEnsemble :: forall n. KnownNat n => DeviceEnsemble n -> IO (Ensemble n)
downloadEnsemble (DeviceEnsemble b dq dp) = do
  let cnt = b * fromIntegral (natVal (Proxy @n))
  q <- newOut cnt; p <- newOut cnt
  VSM.unsafeWith q $ \pq -> VSM.unsafeWith p $ \pp -> withForeignPtr dq $ \a -> withForeignPtr dp $ \c -> do
    c_memcpy pq a (fromIntegral (8 * cnt)) 1 >>= check "download"   -- HAMK_COPY_D2H
    c_memcpy pp c (fromIntegral (8 * cnt)) 1 >>= check "download"
  Ensemble b <$> VS.unsafeFreeze q <*> VS.unsafeFreeze p

-- | Initial conditions of trajectories @first .. first + b - 1@ of an ensemble, drawn ON THE DEVICE from the global
--   trajectory index (@hamk_sample_batch@: per-index splitmix64, uniform boxes @(lo, hi)@ per coordinate for positions
--   and velocities), then 'toPhase' there: a shard of a multi-GPU run needs no host array and no scatter, and every
--   shard layout draws the same bits.  No reference counterpart (one trajectory from a CLI Config, Examples.hs:230-359).
sampleEnsembleDevice :: forall m n. KnownNat n => HipSystem m n -> [(Double, Double)] -> [(Double, Double)] -> Int -> Int -> Word64
                     -> IO (DeviceEnsemble n)
sampleEnsembleDevice (HipSystem h) qBox qdBox first b seed = do
  let n = fromIntegral (natVal (Proxy @n)) :: Int
  when (length qBox /= n || length qdBox /= n) $ throwIO (userError "sampleEnsembleDevice: the boxes need n (lo, hi) pairs")
  dq <- deviceArray (n * b); dv <- deviceArray (n * b); dp <- deviceArray (n * b)
  withForeignPtr h $ \s -> withForeignPtr dq $ \pq -> withForeignPtr dv $ \pv -> withForeignPtr dp $ \pp ->
    withArray (map fst qBox) $ \ql -> withArray (map snd qBox) $ \qh -> withArray (map fst qdBox) $ \vl -> withArray (map snd qdBox) $ \vh -> do
      c_sample_batch s (fromIntegral b) (fromIntegral first) seed ql qh vl vh pq pv memDevice >>= check "sampleEnsemble"
      c_to_phase s (fromIntegral b) pq pv pp memDevice >>= check "toPhase"
      c_synchronize s >>= check "synchronize"
  return (DeviceEnsemble b dq dp)

-- | State the size of the WHOLE ensemble this system's launches are pieces of (@hamk_system_set_ensemble_size@): every
--   shard is then computed by the mapping the library picks for the whole, and any shard layout -- any number of GPUs, a
--   run resumed with another -- reproduces the one-launch bits.  0: back to the per-launch choice.
setEnsembleSize :: HipSystem m n -> Int -> IO ()
setEnsembleSize (HipSystem h) total = withForeignPtr h $ \s ->
  c_system_set_ensemble_size s (fromIntegral total) >>= check "setEnsembleSize"

-- | k classic RK4 steps, in place in HBM (the BASELINE.json hot loop).
rk4StepsDevice :: forall m n. KnownNat n => Double -> Int -> HipSystem m n -> DeviceEnsemble n -> IO ()
rk4StepsDevice dt k (HipSystem h) (DeviceEnsemble b dq dp) =
  withForeignPtr h $ \s -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp ->
    c_rk4_steps s (fromIntegral b) pq pp dt (fromIntegral k) nullPtr memDevice >>= check "rk4Steps"

-- | 'stepHam' (Hamilton.hs:390-402) for every member, in place in HBM.
stepHamDevice :: forall m n. KnownNat n => Double -> HipSystem m n -> DeviceEnsemble n -> IO ()
stepHamDevice r (HipSystem h) (DeviceEnsemble b dq dp) =
  withForeignPtr h $ \s -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp ->
    c_step_ham s (fromIntegral b) pq pp r nullPtr nullPtr memDevice >>= check "stepHam"

-- | @k@ consecutive 'stepHam' calls for every member, in place in HBM, one launch.
iterateStepHamDevice :: forall m n. KnownNat n => Double -> Int -> HipSystem m n -> DeviceEnsemble n -> IO ()
iterateStepHamDevice r k (HipSystem h) (DeviceEnsemble b dq dp) =
  withForeignPtr h $ \s -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp ->
    c_step_ham_iterate s (fromIntegral b) pq pp r (fromIntegral k) 0 nullPtr nullPtr nullPtr nullPtr memDevice >>= check "iterateStepHam"

synchronize :: HipSystem m n -> IO ()
synchronize (HipSystem h) = withForeignPtr h $ \s -> c_synchronize s >>= check "synchronize"

-- | Final gather of the shards of one ensemble (each possibly on another GPU) into one host
--   ensemble, trajectories in shard order.  The only inter-device traffic of the path.
gatherEnsembles :: forall n. KnownNat n => [DeviceEnsemble n] -> IO (Ensemble n)
gatherEnsembles parts = do
  let n = fromIntegral (natVal (Proxy @n)) :: Int
      total = sum (map devSize parts)
      one sel out = withMany withForeignPtr (map sel parts) $ \ptrs ->
        withArrayLen (map (fromIntegral . devSize) parts) $ \g bs -> withArray ptrs $ \pa ->
          VSM.unsafeWith out $ \po -> c_gather (fromIntegral g) (fromIntegral n) bs pa po memHost >>= check "gather"
  q <- newOut (n * total); p <- newOut (n * total)
  one devPositions q; one devMomenta p
  Ensemble total <$> VS.unsafeFreeze q <*> VS.unsafeFreeze p

-- ---------------------------------------------------------------------------
-- one process per GPU (mpirun and the like): the same final gather over RCCL
-- ---------------------------------------------------------------------------
-- | An RCCL communicator bound to the device that was current at 'commCreate' (hamk.h:
--   @hamk_comm_*@).  'gatherEnsembles' serves ONE process that drives every GPU; a process
--   started once per GPU has no peer pointers to hand over.
data Comm = Comm { commPtr :: Ptr HamkComm, commWorld :: Int, commRank :: Int }
-- | The communicator's id: 128 opaque bytes.  Rank 0 draws it and ships it to the other
--   processes by whatever channel the launcher offers (a file, a socket, @MPI_Bcast@).
newtype CommId = CommId [Word8]

commIdBytes :: Int
commIdBytes = 128

commUniqueId :: IO CommId
commUniqueId = allocaArray commIdBytes $ \p -> do
  c_comm_unique_id p >>= check "commUniqueId"
  CommId <$> peekArray commIdBytes p

-- | Collective: returns when all @world@ ranks have called it with the same id.  'setDevice' first.
commCreate :: CommId -> Int -> Int -> IO Comm
commCreate (CommId bytes) world rank = withArrayLen bytes $ \len p -> do
  when (len /= commIdBytes) $ throwIO (ErrorCall "commCreate: the id is 128 bytes")
  alloca $ \out -> do
    c_comm_create p (fromIntegral world) (fromIntegral rank) out >>= check "commCreate"
    c <- peek out
    pure (Comm c world rank)

commDestroy :: Comm -> IO ()
commDestroy (Comm c _ _) = c_comm_destroy c >>= check "commDestroy"

-- | Collective: every rank's shard, rank order, on this rank's device.  @sizes@ holds the
--   members of every rank's shard (the same list on all ranks; ragged shards are fine).
--   'synchronize' the system that advanced the shard first.
allGatherEnsemble :: forall n. KnownNat n => Comm -> [Int] -> DeviceEnsemble n -> IO (DeviceEnsemble n)
allGatherEnsemble (Comm c world rank) sizes mine@(DeviceEnsemble b dq dp) = do
  when (length sizes /= world || sizes !! rank /= b) $
    throwIO (ErrorCall "allGatherEnsemble: one size per rank, sizes !! rank = the shard's")
  let n = fromIntegral (natVal (Proxy @n)) :: Int
      total = sum sizes
  oq <- deviceArray (n * total); op <- deviceArray (n * total)
  withArray (map fromIntegral sizes) $ \bs -> do
    withForeignPtr dq $ \src -> withForeignPtr oq $ \dst -> c_comm_allgather c (fromIntegral n) bs src dst >>= check "allGather q"
    withForeignPtr dp $ \src -> withForeignPtr op $ \dst -> c_comm_allgather c (fromIntegral n) bs src dst >>= check "allGather p"
  touchEnsemble mine
  pure (DeviceEnsemble total oq op)
  where touchEnsemble (DeviceEnsemble _ a b') = touchForeignPtr a >> touchForeignPtr b'

-- ---------------------------------------------------------------------------
-- which binding of hmatrix-gsl's gsl-ode.c the adaptive stepper reproduces
-- ---------------------------------------------------------------------------
-- | hmatrix-gsl's @gsl-ode.c@ carries two bindings: @gsl_odeiv2@ through
--   @gsl_odeiv2_driver_apply@ (its default build, and this library's default) and the old
--   @gsl_odeiv@ loop (@-DGSLODE1@).  They differ from the second output time of 'evolveHam' on
--   (hamk.h: @hamk_system_set_gsl_api@).
data GslApi = GslOdeiv1 | GslOdeiv2 deriving (Eq, Show)

setGslApi :: HipSystem m n -> GslApi -> IO ()
setGslApi (HipSystem h) api = withForeignPtr h $ \s ->
  c_set_gsl_api s (if api == GslOdeiv1 then 1 else 2) >>= check "setGslApi"

-- ---------------------------------------------------------------------------
-- fixed-step launches that check their own invariant; checkpoints
-- ---------------------------------------------------------------------------
-- | HAMK_ST_DRIFT: the launch lost more than the given fraction of its energy.
statusDrift :: Int32
statusDrift = 16

-- | As 'rk4StepsDevice', the launch checking H at entry and exit against @driftTol@; returns the
--   per-trajectory status words (bit 'statusDrift' where the invariant was lost: a fixed step through
--   a close encounter -- the reference's analogous failure raises out of @inv@, Hamilton.hs:321,381).
rk4StepsCheckedDevice :: forall m n. KnownNat n => Double -> Int -> Double -> HipSystem m n -> DeviceEnsemble n -> IO (VS.Vector Int32)
rk4StepsCheckedDevice dt k driftTol sys@(HipSystem h) (DeviceEnsemble b dq dp) = do
  st <- VSM.new b
  withForeignPtr h $ \s -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp -> alloca $ \pd -> do
    c_device_malloc pd (fromIntegral (4 * b)) >>= check "deviceMalloc"
    dst <- peek pd
    c_rk4_steps_checked s (fromIntegral b) pq pp dt (fromIntegral k) driftTol (castPtr dst) memDevice >>= check "rk4StepsChecked"
    synchronize sys
    VSM.unsafeWith st $ \ps -> c_memcpy (castPtr ps) dst (fromIntegral (4 * b)) 1 >>= check "download"
    finalizeForeignPtr =<< newForeignPtr p_device_free dst
  VS.unsafeFreeze st

data CheckpointInfo = CheckpointInfo { ckN :: !Int, ckSize :: !Int, ckStepsDone :: !Int, ckSeed :: !Word64, ckTime :: !Double }
  deriving Show

-- | Dump a device-resident ensemble (plus the caller's bookkeeping) to one flat file; written aside
--   and renamed.  A resumed run continues bit-identically (every kernel is a pure function of the state).
saveCheckpoint :: forall m n. KnownNat n => FilePath -> HipSystem m n -> DeviceEnsemble n -> Int -> Word64 -> Double -> IO ()
saveCheckpoint path sys (DeviceEnsemble b dq dp) stepsDone seed t = do
  synchronize sys
  withCString path $ \cp -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp ->
    c_ck_write cp (fromIntegral (natVal (Proxy @n))) (fromIntegral b) pq pp memDevice (fromIntegral stepsDone) seed t
      >>= check "saveCheckpoint"

checkpointInfo :: FilePath -> IO CheckpointInfo
checkpointInfo path = withCString path $ \cp -> alloca $ \pn -> alloca $ \pb -> alloca $ \ps -> alloca $ \pz -> alloca $ \pt -> do
  c_ck_info cp pn pb ps pz pt >>= check "checkpointInfo"
  CheckpointInfo <$> (fromIntegral <$> peek pn) <*> (fromIntegral <$> peek pb) <*> (fromIntegral <$> peek ps) <*> peek pz <*> peek pt

loadCheckpoint :: forall n. KnownNat n => FilePath -> IO (DeviceEnsemble n, CheckpointInfo)
loadCheckpoint path = do
  info <- checkpointInfo path
  when (ckN info /= fromIntegral (natVal (Proxy @n))) $ throwIO (ErrorCall "loadCheckpoint: the file holds a system of another size")
  let cnt = ckN info * ckSize info
  dq <- deviceArray cnt; dp <- deviceArray cnt
  withCString path $ \cp -> withForeignPtr dq $ \pq -> withForeignPtr dp $ \pp ->
    c_ck_read cp (fromIntegral (ckN info)) (fromIntegral (ckSize info)) pq pp memDevice >>= check "loadCheckpoint"
  return (DeviceEnsemble (ckSize info) dq dp, info)
