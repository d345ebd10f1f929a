// HipTrace.scala — JVM binding of libhiptrace.so for the reference renderer
// (Ivoah/RayTracing, jvm module).  Java FFM / Panama API, JDK >= 22, Scala 3.
//
// NOT COMPILED in the repository that ships this file: its build image has no JDK, scalac or
// sbt.  What can be checked without them is checked there: every byte offset in `Layout`
// below is compared with offsetof() of include/hiptrace.h by a C program, and a GPU test
// fills the three structs through exactly these offsets and renders with them.
//
// Where it goes: jvm/src/main/scala/HipTrace.scala.  The two callers of renderFrame
// (jvm/src/main/scala/main.scala:110 in cmd, :180 in RenderThread.run) then call
// HipTrace.renderFrame with the same arguments; shared/ (the Scala.js and CPU code) is not
// touched.  HipTrace.renderFrame has the signature and the callbacks of
// renderFrame (jvm/src/main/scala/main.scala:73-76).
import java.awt.image.BufferedImage
import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import java.lang.invoke.MethodHandle
import java.util.IdentityHashMap
import scala.collection.mutable.ArrayBuffer

object HipTrace {
  /** Byte offsets of include/hiptrace.h on an LP64 target (x86-64, AArch64). */
  object Layout {
    // struct ht_scene_desc
    val DESC_SIZE = 168L
    val DESC_N_PRIMS = 0L
    val DESC_PRIM_TYPE = 8L
    val DESC_PRIM_MATERIAL = 16L
    val DESC_PRIM_DATA = 24L
    val DESC_PRIM_GROUP = 32L
    val DESC_N_MATERIALS = 40L
    val DESC_MAT_TYPE = 48L
    val DESC_MAT_TEXTURE = 56L
    val DESC_MAT_PARAM = 64L
    val DESC_N_TEXTURES = 72L
    val DESC_TEX_TYPE = 80L
    val DESC_TEX_DATA = 88L
    val DESC_TEX_CHILD = 96L
    val DESC_N_IMAGES = 104L
    val DESC_IMAGE_WIDTH = 112L
    val DESC_IMAGE_HEIGHT = 120L
    val DESC_IMAGE_OFFSET = 128L
    val DESC_IMAGE_TEXELS = 136L
    val DESC_N_PERLIN = 144L
    val DESC_PERLIN_RANVEC = 152L
    val DESC_PERLIN_PERM = 160L
    // struct ht_camera: origin[3] target[3] vup[3] vfov aspect_ratio aperture focus_distance background[3]
    val CAMERA_SIZE = 128L
    val CAMERA_ORIGIN = 0L
    val CAMERA_TARGET = 24L
    val CAMERA_VUP = 48L
    val CAMERA_VFOV = 72L
    val CAMERA_ASPECT_RATIO = 80L
    val CAMERA_APERTURE = 88L
    val CAMERA_FOCUS_DISTANCE = 96L
    val CAMERA_BACKGROUND = 104L
    // struct ht_render_params
    val PARAMS_SIZE = 64L
    val PARAMS_WIDTH = 0L
    val PARAMS_HEIGHT = 4L
    val PARAMS_SAMPLES = 8L
    val PARAMS_SAMPLE_OFFSET = 12L
    val PARAMS_MAX_DEPTH = 16L
    val PARAMS_JITTER = 20L
    val PARAMS_PRECISION = 24L
    val PARAMS_BATCH = 28L
    val PARAMS_SEED = 32L
    val PARAMS_DEVICE = 40L
    val PARAMS_COLLECT_COUNTERS = 44L
    val PARAMS_TILE_FIRST = 48L      // tile sharding: 0 / 0 = every tile of the frame
    val PARAMS_TILE_STRIDE = 52L
    val PARAMS_SHARD_MODE = 56L      // ht_render_sharded only
    val PARAMS_RESERVED = 60L
    // type tags and strides
    val PRIM_STRIDE = 9   // doubles per primitive
    val TEX_STRIDE = 4    // doubles per texture
    val PRIM_SPHERE = 0; val PRIM_XYRECT = 1; val PRIM_XZRECT = 2; val PRIM_YZRECT = 3; val PRIM_TRIANGLE = 4
    val MAT_DIFFUSE = 0; val MAT_GLOSSY = 1; val MAT_GLASS = 2; val MAT_EMISSION = 3
    val TEX_SOLID = 0; val TEX_CHECKER = 1; val TEX_PERLIN = 2; val TEX_IMAGE = 3
    val E_CANCELLED = -5
  }
  import Layout.*

  private val linker = Linker.nativeLinker()
  private val lib = SymbolLookup.libraryLookup(
    Option(System.getProperty("hiptrace.library")).getOrElse("libhiptrace.so"), Arena.global())
  private def fn(name: String, ret: MemoryLayout, args: MemoryLayout*): MethodHandle =
    linker.downcallHandle(lib.find(name).get, FunctionDescriptor.of(ret, args*))
  private def fnVoid(name: String, args: MemoryLayout*): MethodHandle =
    linker.downcallHandle(lib.find(name).get, FunctionDescriptor.ofVoid(args*))

  private val sceneCreate  = fn("ht_scene_create", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS)
  private val sceneDestroy = fnVoid("ht_scene_destroy", ADDRESS)
  private val frameCreate  = fn("ht_frame_create", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS)
  private val frameDestroy = fnVoid("ht_frame_destroy", ADDRESS)
  private val frameAccum   = fn("ht_frame_accumulate", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS)
  private val frameRead    = fn("ht_frame_read", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
  private val cancel       = fn("ht_cancel", JAVA_INT, ADDRESS)
  private val lastError    = fn("ht_last_error", ADDRESS)

  final class HipTraceException(val code: Int, message: String) extends RuntimeException(message)

  private def check(rc: Int): Unit =
    if (rc != 0) {
      val text = lastError.invoke().asInstanceOf[MemorySegment].reinterpret(512).getString(0)
      throw HipTraceException(rc, s"libhiptrace error $rc: $text")
    }

  /** The world as the arrays of ht_scene_desc.  Primitive ids are the order of this walk:
    * the leaves of the BVH that loadScene built, left to right (main.scala:60, BVH.scala:18-33).
    * The closest hit does not depend on that order. */
  final class FlatWorld {
    val primType = ArrayBuffer[Int]()
    val primMaterial = ArrayBuffer[Int]()
    val primData = ArrayBuffer[Double]()
    val matType = ArrayBuffer[Int]()
    val matTexture = ArrayBuffer[Int]()
    val matParam = ArrayBuffer[Double]()
    val texType = ArrayBuffer[Int]()
    val texData = ArrayBuffer[Double]()
    val texChild = ArrayBuffer[Int]()
    val imageWidth = ArrayBuffer[Int]()
    val imageHeight = ArrayBuffer[Int]()
    val imageOffset = ArrayBuffer[Long]()
    val imageTexels = ArrayBuffer[Int]()
    val perlinRanvec = ArrayBuffer[Double]()
    val perlinPerm = ArrayBuffer[Int]()
    // Materials and textures are told apart by identity, not by ==: two Perlin(scale) instances
    // are equal as case classes and still carry different random tables (Texture.scala:38-42).
    private val materialIndex = IdentityHashMap[Material, Integer]()
    private val textureIndex = IdentityHashMap[Texture, Integer]()

    private def texture(t: Texture): Int = {
      val known = textureIndex.get(t)
      if (known != null) return known
      // children first, so that a Checker's record can name them (Texture.scala:20-25)
      val (kind, data, c0, c1) = t match {
        case SolidColor(c) => (TEX_SOLID, Seq(c.x, c.y, c.z, 0.0), 0, 0)
        case Checker(size, t1, t2) => (TEX_CHECKER, Seq(size, 0.0, 0.0, 0.0), texture(t1), texture(t2))
        case p: Perlin =>
          val table = perlinRanvec.length / (256 * 3)
          p.ranvec.foreach { v => perlinRanvec += v.x; perlinRanvec += v.y; perlinRanvec += v.z }
          perlinPerm ++= p.perm_x; perlinPerm ++= p.perm_y; perlinPerm ++= p.perm_z
          (TEX_PERLIN, Seq(p.scale, 0.0, 0.0, 0.0), table, 0)
        case Image(width, height, pixels) =>
          val image = imageWidth.length
          imageWidth += width; imageHeight += height; imageOffset += imageTexels.length.toLong
          // the 0x00RRGGBB word Vec3.fromRGB was given (Vec3.scala:36-40): c = k/255.0 exactly
          pixels.foreach { c =>
            imageTexels += ((c.x * 255).round.toInt << 16) | ((c.y * 255).round.toInt << 8) | (c.z * 255).round.toInt
          }
          (TEX_IMAGE, Seq(0.0, 0.0, 0.0, 0.0), image, 0)
      }
      val index = texType.length
      texType += kind; texData ++= data; texChild += c0; texChild += c1
      textureIndex.put(t, index)
      index
    }

    private def material(m: Material): Int = {
      val known = materialIndex.get(m)
      if (known != null) return known
      val (kind, tex, param) = m match {
        case Diffuse(t) => (MAT_DIFFUSE, texture(t), 0.0)          // Material.scala:9
        case Glossy(t, roughness) => (MAT_GLOSSY, texture(t), roughness)  // Material.scala:15
        case Glass(t, ior) => (MAT_GLASS, texture(t), ior)         // Material.scala:23
        case Emission(t, strength) => (MAT_EMISSION, texture(t), strength)  // Material.scala:52
      }
      val index = matType.length
      matType += kind; matTexture += tex; matParam += param
      materialIndex.put(m, index)
      index
    }

    private def prim(kind: Int, m: Material, values: Double*): Unit = {
      primType += kind
      primMaterial += material(m)
      primData ++= values
      primData ++= Seq.fill(PRIM_STRIDE - values.length)(0.0)
    }

    /** Hittable.scala:25-115, BVH.scala:4.  A one-object world is BVH(only, only) (BVH.scala:30):
      * the same object twice, flattened once. */
    def add(h: Hittable): Unit = h match {
      case BVH(left, right) => add(left); if (right ne left) add(right)
      case Sphere(c, radius, m) => prim(PRIM_SPHERE, m, c.x, c.y, c.z, radius)
      case XYRect(x0, x1, y0, y1, z, m) => prim(PRIM_XYRECT, m, x0, x1, y0, y1, z)
      case XZRect(x0, x1, z0, z1, y, m) => prim(PRIM_XZRECT, m, x0, x1, z0, z1, y)
      case YZRect(y0, y1, z0, z1, x, m) => prim(PRIM_YZRECT, m, y0, y1, z0, z1, x)
      case Triangle((a, b, c), m) => prim(PRIM_TRIANGLE, m, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z)
      case other => throw IllegalArgumentException(s"HipTrace cannot flatten $other")
    }

    /** The arrays as native memory and the ht_scene_desc that points at them. */
    def toNative(arena: Arena): MemorySegment = {
      def ints(a: ArrayBuffer[Int]) = if (a.isEmpty) MemorySegment.NULL else arena.allocateFrom(JAVA_INT, a.toArray*)
      def longs(a: ArrayBuffer[Long]) = if (a.isEmpty) MemorySegment.NULL else arena.allocateFrom(JAVA_LONG, a.toArray*)
      def doubles(a: ArrayBuffer[Double]) = if (a.isEmpty) MemorySegment.NULL else arena.allocateFrom(JAVA_DOUBLE, a.toArray*)
      val d = arena.allocate(DESC_SIZE, 8)
      d.fill(0.toByte)
      d.set(JAVA_INT, DESC_N_PRIMS, primType.length)
      d.set(ADDRESS, DESC_PRIM_TYPE, ints(primType))
      d.set(ADDRESS, DESC_PRIM_MATERIAL, ints(primMaterial))
      d.set(ADDRESS, DESC_PRIM_DATA, doubles(primData))
      d.set(ADDRESS, DESC_PRIM_GROUP, MemorySegment.NULL)
      d.set(JAVA_INT, DESC_N_MATERIALS, matType.length)
      d.set(ADDRESS, DESC_MAT_TYPE, ints(matType))
      d.set(ADDRESS, DESC_MAT_TEXTURE, ints(matTexture))
      d.set(ADDRESS, DESC_MAT_PARAM, doubles(matParam))
      d.set(JAVA_INT, DESC_N_TEXTURES, texType.length)
      d.set(ADDRESS, DESC_TEX_TYPE, ints(texType))
      d.set(ADDRESS, DESC_TEX_DATA, doubles(texData))
      d.set(ADDRESS, DESC_TEX_CHILD, ints(texChild))
      d.set(JAVA_INT, DESC_N_IMAGES, imageWidth.length)
      d.set(ADDRESS, DESC_IMAGE_WIDTH, ints(imageWidth))
      d.set(ADDRESS, DESC_IMAGE_HEIGHT, ints(imageHeight))
      d.set(ADDRESS, DESC_IMAGE_OFFSET, longs(imageOffset))
      d.set(ADDRESS, DESC_IMAGE_TEXELS, ints(imageTexels))
      d.set(JAVA_INT, DESC_N_PERLIN, perlinRanvec.length / (256 * 3))
      d.set(ADDRESS, DESC_PERLIN_RANVEC, doubles(perlinRanvec))
      d.set(ADDRESS, DESC_PERLIN_PERM, ints(perlinPerm))
      d
    }
  }

  object Flatten {
    def apply(world: Hittable): FlatWorld = { val f = FlatWorld(); f.add(world); f }
  }

  // One scene handle is kept for the world it was made from: the GUI renders the same world
  // again and again (main.scala:180), and flattening + BVH + upload need not be repeated.
  private var cachedWorld: Hittable = null
  private var cachedScene: MemorySegment = MemorySegment.NULL

  private def sceneFor(world: Hittable): MemorySegment = synchronized {
    if (!(world eq cachedWorld)) {
      if (cachedScene != MemorySegment.NULL) sceneDestroy.invoke(cachedScene)
      cachedWorld = null
      cachedScene = MemorySegment.NULL
      val arena = Arena.ofConfined()   // the description is copied by ht_scene_create
      try {
        val desc = Flatten(world).toNative(arena)
        val out = arena.allocate(ADDRESS)
        check(sceneCreate.invoke(desc, -1, out).asInstanceOf[Int])
        cachedScene = out.get(ADDRESS, 0)
        cachedWorld = world
      } finally arena.close()
    }
    cachedScene
  }

  private def nativeCamera(arena: Arena, c: Camera): MemorySegment = {
    val m = arena.allocate(CAMERA_SIZE, 8)
    def vec(offset: Long, v: Vec3): Unit = {
      m.set(JAVA_DOUBLE, offset, v.x); m.set(JAVA_DOUBLE, offset + 8, v.y); m.set(JAVA_DOUBLE, offset + 16, v.z)
    }
    vec(CAMERA_ORIGIN, c.origin)
    vec(CAMERA_TARGET, c.target)
    vec(CAMERA_VUP, c.vup)
    m.set(JAVA_DOUBLE, CAMERA_VFOV, c.vfov)
    m.set(JAVA_DOUBLE, CAMERA_ASPECT_RATIO, c.aspect_ratio)
    m.set(JAVA_DOUBLE, CAMERA_APERTURE, c.aperture)
    m.set(JAVA_DOUBLE, CAMERA_FOCUS_DISTANCE, c.focus_distance)
    vec(CAMERA_BACKGROUND, c.background)
    m
  }

  /** Samples per pixel between two looks at shouldBreak / two updates of the picture. */
  var batch: Int = 64
  /** Philox key.  The reference's RNG is unseeded (Vec3.scala:10-34); any fixed key is as good. */
  var seed: Long = 0L

  /** Same signature and callbacks as renderFrame (jvm/src/main/scala/main.scala:73-76). */
  def renderFrame(camera: Camera, world: Hittable, img: BufferedImage, samples: Int,
                  update: Int => Unit = _ => (), finish: Double => Unit = _ => (),
                  shouldBreak: () => Boolean = () => false): Unit = {
    val start = System.currentTimeMillis()
    val (w, h) = (img.getWidth, img.getHeight)
    val scene = sceneFor(world)
    val arena = Arena.ofConfined()
    var frame = MemorySegment.NULL
    try {
      val out = arena.allocate(ADDRESS)
      check(frameCreate.invoke(scene, w, h, out).asInstanceOf[Int])
      frame = out.get(ADDRESS, 0)
      val cam = nativeCamera(arena, camera)
      val packed = arena.allocate(JAVA_INT, w.toLong * h)
      val params = arena.allocate(PARAMS_SIZE, 8)
      params.fill(0.toByte)
      params.set(JAVA_INT, PARAMS_WIDTH, w)
      params.set(JAVA_INT, PARAMS_HEIGHT, h)
      params.set(JAVA_INT, PARAMS_MAX_DEPTH, 50)           // Camera.scala:25
      params.set(JAVA_INT, PARAMS_JITTER, 1)
      params.set(JAVA_INT, PARAMS_PRECISION, 0)            // fp32 device arithmetic
      params.set(JAVA_INT, PARAMS_BATCH, 0)                // the library's launch size
      params.set(JAVA_LONG, PARAMS_SEED, seed)
      params.set(JAVA_INT, PARAMS_DEVICE, -1)              // the scene's device
      params.set(JAVA_INT, PARAMS_COLLECT_COUNTERS, 0)
      var done = 0
      while (done < samples && !shouldBreak()) {           // main.scala:80
        val n = math.min(batch, samples - done)
        params.set(JAVA_INT, PARAMS_SAMPLES, n)
        params.set(JAVA_INT, PARAMS_SAMPLE_OFFSET, done)
        check(frameAccum.invoke(frame, cam, params, MemorySegment.NULL).asInstanceOf[Int])
        check(frameRead.invoke(frame, MemorySegment.NULL, packed).asInstanceOf[Int])
        // rows arrive top first, as img.setRGB(i, H-j-1, ..) leaves them (main.scala:93)
        img.setRGB(0, 0, w, h, packed.toArray(JAVA_INT), 0, w)
        done += n
        // all rows advance together on the GPU: report the row the reference's loop would
        // have reached after the same share of the work (main.scala:95)
        update(math.max(0, (h.toLong * done / samples).toInt - 1))
      }
    } finally {
      if (frame != MemorySegment.NULL) frameDestroy.invoke(frame)
      arena.close()
    }
    finish((System.currentTimeMillis() - start) / 1000.0)  // main.scala:97-98
  }

  /** Stops a frame that another thread is rendering at its next launch boundary (the GUI's
    * Stop item, main.scala:254-263, besides shouldBreak). */
  def stop(): Unit = synchronized {
    if (cachedScene != MemorySegment.NULL) cancel.invoke(cachedScene).asInstanceOf[Int]
  }
}
