{-# LANGUAGE BangPatterns #-}
{-# LANGUAGE ForeignFunctionInterface #-}

-- | Binding of the MI355X render core (librtcore.so, include/rtcore.h) for omeldar/raytracer.
--
-- Drop this file in as app/Rendering/GpuRender.hs, apply integration/ImageGenerator.patch and
-- integration/raytracer.cabal.patch.  `renderOnGpu` replaces the worker fan-out of `createPPM`
-- (app/Rendering/ImageGenerator.hs:79-94, 126): it walks the `BVHNode` tree that
-- `parseSceneObjects` built into the flat arrays of `rt_scene_desc`, hands materials, lights,
-- background, sky and camera over, renders ONE frame with one blocking call on every visible GPU
-- (`rt_init_devices 0 nullPtr`) and returns the 8-bit frame in file order (row 0 = top).
--
-- GHC is not part of the image this was written in, so the module has never been compiled; what IS
-- checked mechanically (tests/test_haskell_offsets.py) is every struct size and field offset below
-- against `offsetof` / `sizeof` of include/rtcore.h, the list of imported C symbols against the
-- library's exports, and their arities against the header.
--
-- Typeable: `SomeHittable` (app/Hittable/Wrapped.hs:8) already demands `Typeable a`, and since GHC 7.10
-- every data type has a Typeable instance without a deriving clause, so `cast` below works for Sphere
-- and Plane (which derive only Show, Sphere.hs:17 / Plane.hs:17) exactly as `extractTriangles`
-- (app/Rendering/SceneParser.hs:98-103) uses it for Triangle.  base >= 4.17 (raytracer.cabal:58) is GHC 9.4+.
module Rendering.GpuRender
  ( renderOnGpu,
    frameRows,
  )
where

import Codec.Picture (Image (..))
import Config
  ( BackgroundSettings (..),
    Config (..),
    ImageSettings (..),
    RaytracerSettings (..),
  )
import Control.Monad (forM_, when)
import Core.Vec3 (Vec3 (..))
import qualified Data.Map as M
import Data.Typeable (cast)
import qualified Data.Vector as DV
import qualified Data.Vector.Storable as SV
import qualified Data.Vector.Storable.Mutable as SVM
import Data.Word (Word32, Word64, Word8)
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types (CInt (..))
import Foreign.Marshal.Alloc (alloca, allocaBytes)
import Foreign.Marshal.Utils (fillBytes)
import Foreign.Ptr (Ptr, castPtr, nullPtr, plusPtr)
import Foreign.Storable (peek, pokeByteOff)
import Data.Int (Int32)
import Hittable.BVH (BVHNode (..))
import Hittable.BoundingBox (AABB (..))
import Hittable.HittableList (SomeHittable (..))
import qualified Hittable.Objects.Plane as P
import qualified Hittable.Objects.Sphere as S
import qualified Hittable.Objects.Triangle as T
import qualified Rendering.Camera as Cam
import qualified Rendering.Light as L
import Rendering.Material (Material (..))
import Rendering.SkySphere (SkySphere (..))
import System.Random (randomIO)

-- ---------------------------------------------------------------------------------------------
-- C ABI (include/rtcore.h).  `safe`: the call blocks for the whole frame, no callbacks.
-- ---------------------------------------------------------------------------------------------
data RtScene

foreign import ccall safe "rt_init_devices" c_rt_init_devices :: Int32 -> Ptr Int32 -> IO CInt

foreign import ccall safe "rt_scene_create" c_rt_scene_create :: Ptr () -> Ptr (Ptr RtScene) -> IO CInt

foreign import ccall safe "rt_render" c_rt_render :: Ptr RtScene -> Ptr () -> Ptr () -> Ptr () -> Ptr () -> IO CInt

foreign import ccall safe "rt_scene_destroy" c_rt_scene_destroy :: Ptr RtScene -> IO ()

foreign import ccall safe "rt_last_error" c_rt_last_error :: IO CString

foreign import ccall safe "rt_abi_version" c_rt_abi_version :: IO CInt

-- BEGIN-ABI-LAYOUT  (x86-64 SysV; generated from tools/abi_layout.c, verified by tests/test_haskell_offsets.py)
rt_abi_version_expected :: Int
rt_abi_version_expected = 4

sizeof_rt_material, off_rt_material_id, off_rt_material_flags, off_rt_material_diffuse, off_rt_material_emission, off_rt_material_shininess, off_rt_material_ior, off_rt_material_transmission :: Int
sizeof_rt_material = 80
off_rt_material_id = 0
off_rt_material_flags = 4
off_rt_material_diffuse = 8
off_rt_material_emission = 32
off_rt_material_shininess = 56
off_rt_material_ior = 64
off_rt_material_transmission = 72

sizeof_rt_light, off_rt_light_kind, off_rt_light_reserved, off_rt_light_vec, off_rt_light_intensity :: Int
sizeof_rt_light = 56
off_rt_light_kind = 0
off_rt_light_reserved = 4
off_rt_light_vec = 8
off_rt_light_intensity = 32

sizeof_rt_scene_desc, off_rt_scene_desc_n_nodes, off_rt_scene_desc_node_bounds, off_rt_scene_desc_node_child, off_rt_scene_desc_n_prims, off_rt_scene_desc_prim_kind, off_rt_scene_desc_prim_data, off_rt_scene_desc_prim_material, off_rt_scene_desc_prim_id, off_rt_scene_desc_n_materials, off_rt_scene_desc_materials, off_rt_scene_desc_n_lights, off_rt_scene_desc_lights, off_rt_scene_desc_bg_kind, off_rt_scene_desc_bg_color1, off_rt_scene_desc_bg_color2, off_rt_scene_desc_sky_format, off_rt_scene_desc_sky_width, off_rt_scene_desc_sky_height, off_rt_scene_desc_sky_pixels :: Int
sizeof_rt_scene_desc = 176
off_rt_scene_desc_n_nodes = 0
off_rt_scene_desc_node_bounds = 8
off_rt_scene_desc_node_child = 16
off_rt_scene_desc_n_prims = 24
off_rt_scene_desc_prim_kind = 32
off_rt_scene_desc_prim_data = 40
off_rt_scene_desc_prim_material = 48
off_rt_scene_desc_prim_id = 56
off_rt_scene_desc_n_materials = 64
off_rt_scene_desc_materials = 72
off_rt_scene_desc_n_lights = 80
off_rt_scene_desc_lights = 88
off_rt_scene_desc_bg_kind = 96
off_rt_scene_desc_bg_color1 = 104
off_rt_scene_desc_bg_color2 = 128
off_rt_scene_desc_sky_format = 152
off_rt_scene_desc_sky_width = 156
off_rt_scene_desc_sky_height = 160
off_rt_scene_desc_sky_pixels = 168

sizeof_rt_camera, off_rt_camera_origin, off_rt_camera_lower_left, off_rt_camera_horizontal, off_rt_camera_vertical, off_rt_camera_u, off_rt_camera_v, off_rt_camera_w, off_rt_camera_aperture :: Int
sizeof_rt_camera = 176
off_rt_camera_origin = 0
off_rt_camera_lower_left = 24
off_rt_camera_horizontal = 48
off_rt_camera_vertical = 72
off_rt_camera_u = 96
off_rt_camera_v = 120
off_rt_camera_w = 144
off_rt_camera_aperture = 168

sizeof_rt_params, off_rt_params_width, off_rt_params_height, off_rt_params_samples_per_pixel, off_rt_params_antialiasing, off_rt_params_max_bounces, off_rt_params_rr_start_bounce, off_rt_params_exposure, off_rt_params_gamma, off_rt_params_seed, off_rt_params_collect_stats, off_rt_params_box_precision :: Int
sizeof_rt_params = 56
off_rt_params_width = 0
off_rt_params_height = 4
off_rt_params_samples_per_pixel = 8
off_rt_params_antialiasing = 12
off_rt_params_max_bounces = 16
off_rt_params_rr_start_bounce = 20
off_rt_params_exposure = 24
off_rt_params_gamma = 32
off_rt_params_seed = 40
off_rt_params_collect_stats = 48
off_rt_params_box_precision = 52

sizeof_rt_output, off_rt_output_radiance, off_rt_output_color, off_rt_output_rgb8, off_rt_output_prim_id :: Int
sizeof_rt_output = 32
off_rt_output_radiance = 0
off_rt_output_color = 8
off_rt_output_rgb8 = 16
off_rt_output_prim_id = 24

sizeof_rt_stats :: Int
sizeof_rt_stats = 312
-- END-ABI-LAYOUT

-- constants of include/rtcore.h
rtPrimTriangle, rtPrimSphere, rtPrimPlane, rtLightPoint, rtLightDirectional, rtBgGradient, rtBgSolid, rtSkyNone, rtSkyRgb8, rtSkyRgbf :: Int32
rtPrimTriangle = 0
rtPrimSphere = 1
rtPrimPlane = 2
rtLightPoint = 0
rtLightDirectional = 1
rtBgGradient = 0
rtBgSolid = 1
rtSkyNone = 0
rtSkyRgb8 = 1
rtSkyRgbf = 2

rtMatHasEmission, rtMatHasShininess, rtMatHasIor, rtMatHasTransmission :: Word32
rtMatHasEmission = 1
rtMatHasShininess = 2
rtMatHasIor = 4
rtMatHasTransmission = 8

-- ---------------------------------------------------------------------------------------------
-- BVHNode -> flat arrays (rt_scene_desc, include/rtcore.h:88-115)
-- ---------------------------------------------------------------------------------------------

-- | One node in pre-order: its own box and the two child words
--   (internal: left index, right index; leaf: -1 - firstPrim, primitive count).
data NodeRec = NodeRec !AABB !Int32 !Int32

-- | Pre-order walk of the tree built by `constructBVHWithLimit` (app/Hittable/BVH.hs:29-46).
--   A node's index is the number of nodes emitted before it, so the left child of node i is i + 1
--   and the right child follows the left subtree.  A leaf's primitives are appended in
--   `DV.toList` order: that order decides ties (app/Utils/HitHelpers.hs:11-16).
--   Difference lists keep the walk linear in the size of the tree.
flattenTree :: BVHNode -> ([NodeRec], [SomeHittable])
flattenTree root =
  let (ns, ps, _, _) = go root 0 0
   in (ns [], ps [])
  where
    go :: BVHNode -> Int -> Int -> ([NodeRec] -> [NodeRec], [SomeHittable] -> [SomeHittable], Int, Int)
    go (BVHLeaf objs box) _ firstPrim =
      let n = DV.length objs
       in ((NodeRec box (fromIntegral (-1 - firstPrim)) (fromIntegral n) :), (DV.toList objs ++), 1, n)
    go (BVHInternal l r box) i firstPrim =
      let (nl, pl, cl, kl) = go l (i + 1) firstPrim
          (nr, pr, cr, kr) = go r (i + 1 + cl) (firstPrim + kl)
          self = NodeRec box (fromIntegral (i + 1)) (fromIntegral (i + 1 + cl))
       in ((self :) . nl . nr, pl . pr, 1 + cl + cr, kl + kr)

xyz :: Vec3 -> [Double]
xyz (Vec3 a b c) = [a, b, c]

-- | (kind, 9 doubles, material id) of one scene object (include/rtcore.h:45-48).
primOf :: SomeHittable -> (Int32, [Double], Int32)
primOf (SomeHittable o)
  | Just (T.Triangle a b c _ m) <- cast o = (rtPrimTriangle, xyz a ++ xyz b ++ xyz c, fromIntegral m)
  | Just (S.Sphere c r _ m) <- cast o = (rtPrimSphere, xyz c ++ [r, 0, 0, 0, 0, 0], fromIntegral m)
  | Just (P.Plane p n _ m) <- cast o = (rtPrimPlane, xyz p ++ xyz n ++ [0, 0, 0], fromIntegral m)
  | otherwise = error "GpuRender: a Hittable that is neither Triangle, Sphere nor Plane"

-- ---------------------------------------------------------------------------------------------
-- struct marshalling
-- ---------------------------------------------------------------------------------------------
pokeVec3 :: Ptr a -> Int -> Vec3 -> IO ()
pokeVec3 p off (Vec3 a b c) = do
  pokeByteOff p off a
  pokeByteOff p (off + 8) b
  pokeByteOff p (off + 16) c

-- | One rt_material per entry of the reference's `Map Int Material` (Trace.hs:27-28);
--   the `Maybe` fields of app/Rendering/Material.hs:13-21 become RT_MAT_HAS_* flag bits.
pokeMaterial :: Ptr a -> (Int, Material) -> IO ()
pokeMaterial p (mid, m) = do
  let flag b v = maybe 0 (const b) v
      flags = flag rtMatHasEmission (emissionColor m) + flag rtMatHasShininess (shininess m) + flag rtMatHasIor (ior m) + flag rtMatHasTransmission (transmission m) :: Word32
  pokeByteOff p off_rt_material_id (fromIntegral mid :: Int32)
  pokeByteOff p off_rt_material_flags flags
  pokeVec3 p off_rt_material_diffuse (diffuseColor m)
  pokeVec3 p off_rt_material_emission (maybe (Vec3 0 0 0) id (emissionColor m))
  pokeByteOff p off_rt_material_shininess (maybe 0 id (shininess m) :: Double)
  pokeByteOff p off_rt_material_ior (maybe 0 id (ior m) :: Double)
  pokeByteOff p off_rt_material_transmission (maybe 0 id (transmission m) :: Double)

pokeLight :: Ptr a -> L.Light -> IO ()
pokeLight p (L.PointLight pos i) = do
  pokeByteOff p off_rt_light_kind rtLightPoint
  pokeByteOff p off_rt_light_reserved (0 :: Int32)
  pokeVec3 p off_rt_light_vec pos
  pokeVec3 p off_rt_light_intensity i
pokeLight p (L.DirectionalLight dir i) = do
  pokeByteOff p off_rt_light_kind rtLightDirectional
  pokeByteOff p off_rt_light_reserved (0 :: Int32)
  pokeVec3 p off_rt_light_vec dir
  pokeVec3 p off_rt_light_intensity i

-- | The `Camera` record (app/Rendering/Camera.hs:13-23), field by field.
pokeCamera :: Ptr a -> Cam.Camera -> IO ()
pokeCamera p c = do
  pokeVec3 p off_rt_camera_origin (Cam.origin c)
  pokeVec3 p off_rt_camera_lower_left (Cam.lowerLeftCorner c)
  pokeVec3 p off_rt_camera_horizontal (Cam.horizontal c)
  pokeVec3 p off_rt_camera_vertical (Cam.vertical c)
  pokeVec3 p off_rt_camera_u (Cam.u c)
  pokeVec3 p off_rt_camera_v (Cam.v c)
  pokeVec3 p off_rt_camera_w (Cam.w c)
  pokeByteOff p off_rt_camera_aperture (Cam.aperture c)

-- | Runs an action with the sky texels as a C pointer: JuicyPixels keeps an image as one
--   storable vector of components, row-major, row 0 = top (what `pixelAt` indexes,
--   app/Rendering/SkySphere.hs:42-47).
withSky :: Maybe SkySphere -> (Int32 -> Int32 -> Int32 -> Ptr () -> IO b) -> IO b
withSky Nothing k = k rtSkyNone 0 0 nullPtr
withSky (Just (LDRSkySphere w h img)) k = SV.unsafeWith (imageData img) $ \p -> k rtSkyRgb8 (fromIntegral w) (fromIntegral h) (castPtr p)
withSky (Just (HDRSkySphere w h img)) k = SV.unsafeWith (imageData img) $ \p -> k rtSkyRgbf (fromIntegral w) (fromIntegral h) (castPtr p)

check :: String -> CInt -> IO ()
check what rc = when (rc /= 0) $ do
  msg <- c_rt_last_error >>= peekCString
  error ("GpuRender: " ++ what ++ " failed (" ++ show rc ++ "): " ++ msg) -- like the `error` calls of the reference (SURVEY 5.3)

-- ---------------------------------------------------------------------------------------------
-- the replacement of the worker fan-out
-- ---------------------------------------------------------------------------------------------

-- | Renders one frame on every visible GPU and returns W*H*3 bytes, file order (row 0 = top):
--   `truncate (c * 255.999)` of what `pixelColor` returns (app/Rendering/RenderWorker.hs:76-82, 108-109).
renderOnGpu :: Config -> BVHNode -> M.Map Int Material -> Maybe SkySphere -> Cam.Camera -> [L.Light] -> IO (SV.Vector Word8)
renderOnGpu config bvh materialMap skySphere cameraObj sceneLights = do
  ver <- c_rt_abi_version
  when (fromIntegral ver /= rt_abi_version_expected) $ error ("GpuRender: librtcore ABI version " ++ show ver ++ ", this binding was written for " ++ show rt_abi_version_expected)
  c_rt_init_devices 0 nullPtr >>= check "rt_init_devices" -- 0 / NULL: every visible GPU; tiles are interleaved over them inside the library

  let img = image config
      imgW = width img
      imgH = height img
      (nodeRecs, prims) = flattenTree bvh
      nNodes = length nodeRecs
      nPrims = length prims
      prims' = map primOf prims
      nodeBounds = SV.fromList (concat [xyz lo ++ xyz hi | NodeRec (AABB lo hi) _ _ <- nodeRecs]) :: SV.Vector Double
      nodeChild = SV.fromList (concat [[a, b] | NodeRec _ a b <- nodeRecs]) :: SV.Vector Int32
      primKind = SV.fromList [k | (k, _, _) <- prims'] :: SV.Vector Int32
      primData = SV.fromList (concat [d | (_, d, _) <- prims']) :: SV.Vector Double
      primMat = SV.fromList [m | (_, _, m) <- prims'] :: SV.Vector Int32
      mats = M.toList materialMap
      nMats = length mats
      nLights = length sceneLights

  seed <- randomIO :: IO Word64 -- the reference seeds from system entropy too (app/Utils/RandomHelper.hs:22): a different image every run
  frame <- SVM.new (imgW * imgH * 3) :: IO (SVM.IOVector Word8)

  SV.unsafeWith nodeBounds $ \pBounds ->
    SV.unsafeWith nodeChild $ \pChild ->
      SV.unsafeWith primKind $ \pKind ->
        SV.unsafeWith primData $ \pData ->
          SV.unsafeWith primMat $ \pMat ->
            allocaBytes (max 1 nMats * sizeof_rt_material) $ \pMats ->
              allocaBytes (max 1 nLights * sizeof_rt_light) $ \pLights ->
                withSky skySphere $ \skyFmt skyW skyH pSky ->
                  allocaBytes sizeof_rt_scene_desc $ \desc ->
                    allocaBytes sizeof_rt_camera $ \cam ->
                      allocaBytes sizeof_rt_params $ \params ->
                        allocaBytes sizeof_rt_output $ \out ->
                          allocaBytes sizeof_rt_stats $ \stats ->
                            alloca $ \pScene -> do
                              forM_ (zip [0 ..] mats) $ \(k, m) -> pokeMaterial (pMats `plusPtr` (k * sizeof_rt_material)) m
                              forM_ (zip [0 ..] sceneLights) $ \(k, l) -> pokeLight (pLights `plusPtr` (k * sizeof_rt_light)) l

                              fillBytes desc 0 sizeof_rt_scene_desc
                              pokeByteOff desc off_rt_scene_desc_n_nodes (fromIntegral nNodes :: Int32)
                              pokeByteOff desc off_rt_scene_desc_node_bounds pBounds
                              pokeByteOff desc off_rt_scene_desc_node_child pChild
                              pokeByteOff desc off_rt_scene_desc_n_prims (fromIntegral nPrims :: Int32)
                              pokeByteOff desc off_rt_scene_desc_prim_kind pKind
                              pokeByteOff desc off_rt_scene_desc_prim_data pData
                              pokeByteOff desc off_rt_scene_desc_prim_material pMat
                              pokeByteOff desc off_rt_scene_desc_prim_id (nullPtr :: Ptr Int32) -- NULL: prim_id[k] = k (only reported back)
                              pokeByteOff desc off_rt_scene_desc_n_materials (fromIntegral nMats :: Int32)
                              pokeByteOff desc off_rt_scene_desc_materials pMats
                              pokeByteOff desc off_rt_scene_desc_n_lights (fromIntegral nLights :: Int32)
                              pokeByteOff desc off_rt_scene_desc_lights pLights
                              case background config of
                                Gradient c1 c2 -> do
                                  pokeByteOff desc off_rt_scene_desc_bg_kind rtBgGradient
                                  pokeVec3 desc off_rt_scene_desc_bg_color1 c1
                                  pokeVec3 desc off_rt_scene_desc_bg_color2 c2
                                SolidColor c -> do
                                  pokeByteOff desc off_rt_scene_desc_bg_kind rtBgSolid
                                  pokeVec3 desc off_rt_scene_desc_bg_color1 c
                                  pokeVec3 desc off_rt_scene_desc_bg_color2 c
                              pokeByteOff desc off_rt_scene_desc_sky_format skyFmt
                              pokeByteOff desc off_rt_scene_desc_sky_width skyW
                              pokeByteOff desc off_rt_scene_desc_sky_height skyH
                              pokeByteOff desc off_rt_scene_desc_sky_pixels pSky

                              pokeCamera cam cameraObj

                              fillBytes params 0 sizeof_rt_params
                              pokeByteOff params off_rt_params_width (fromIntegral imgW :: Int32)
                              pokeByteOff params off_rt_params_height (fromIntegral imgH :: Int32)
                              pokeByteOff params off_rt_params_samples_per_pixel (fromIntegral (samplesPerPixel img) :: Int32)
                              pokeByteOff params off_rt_params_antialiasing (if antialiasing img then 1 else 0 :: Int32)
                              pokeByteOff params off_rt_params_max_bounces (fromIntegral (maxBounces (raytracer config)) :: Int32)
                              pokeByteOff params off_rt_params_rr_start_bounce (0 :: Int32) -- Russian roulette is not in the reference (Config.hs:57)
                              pokeByteOff params off_rt_params_exposure (exposure img)
                              pokeByteOff params off_rt_params_gamma (gamma img)
                              pokeByteOff params off_rt_params_seed seed
                              pokeByteOff params off_rt_params_collect_stats (0 :: Int32)
                              pokeByteOff params off_rt_params_box_precision (0 :: Int32) -- 1: walk this very tree literally (step-by-step comparisons)

                              c_rt_scene_create desc pScene >>= check "rt_scene_create"
                              scene <- peek pScene
                              SVM.unsafeWith frame $ \pFrame -> do
                                pokeByteOff out off_rt_output_radiance (nullPtr :: Ptr Double) -- not wanted: the writer needs the 8-bit frame only
                                pokeByteOff out off_rt_output_color (nullPtr :: Ptr Double) -- (ask for this one to keep `showPixel` in Haskell)
                                pokeByteOff out off_rt_output_prim_id (nullPtr :: Ptr Int32)
                                pokeByteOff out off_rt_output_rgb8 pFrame
                                fillBytes stats 0 sizeof_rt_stats
                                rc <- c_rt_render scene cam params out stats
                                c_rt_scene_destroy scene
                                check "rt_render" rc
  SV.unsafeFreeze frame

-- | The frame as the row strings the writer thread of `createPPM` expects (ImageGenerator.hs:101-115):
--   key j = file row j (top row first), value = one "r g b" line per pixel, as `showPixel` + `unlines` produce
--   (app/Rendering/RenderWorker.hs:71, 108-109).
frameRows :: Int -> Int -> SV.Vector Word8 -> M.Map Int String
frameRows imgW imgH frame =
  M.fromList
    [ (j, unlines [unwords [show (frame SV.! (o + c)) | c <- [0, 1, 2]] | i <- [0 .. imgW - 1], let o = 3 * (j * imgW + i)])
      | j <- [0 .. imgH - 1]
    ]
