/* Headless OpenGL rasterisation through Mesa's software driver (llvmpipe / softpipe, swrast_dri.so) WITHOUT an X server, EGL or
 * OSMesa: the DRI "swrast" interface is driven directly with a do-nothing loader, everything is rendered into an FBO.  Build-container
 * tool for tests/golden/make_golden_gl.py: it replays the GL calls of the reference's utils/renderer.py Renderer (:326-451) --
 * GLSL 330 core shaders 'vertex_attribute' / 'position', non-indexed GL_TRIANGLES, mvp uploaded with transpose = GL_TRUE, RGBA32F colour
 * texture + DEPTH24_STENCIL8 renderbuffer, clear, GL_DEPTH_TEST, GL_CULL_FACE, glReadPixels(GL_RGBA, GL_FLOAT) -- on a real OpenGL
 * implementation, which is what pins oracle/raster_oracle.c.
 *
 *   mesa_raster <in.bin> <out.bin>
 *   in : int32 W, H, nverts, vs_len, fs_len; float mvp[16] (row-major); float vertices[nverts*3]; float attrs[nverts*3]; then the GLSL sources of the
 *        vertex and the fragment shader (vs_len, fs_len bytes) -- the caller reads them from the reference's utils/renderer.py at run time
 *        (tests/golden/make_golden_gl.py); no shader text lives in this file
 *   out: float RGBA[H*W*4], rows as glReadPixels returns them (row 0 = bottom)
 * gcc -O1 mesa_raster.c -o mesa_raster -ldl      (needs mesa-common-dev's GL/internal/dri_interface.h and libgl1-mesa-dri) */
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv) { *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv) {}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv) { memset(data, 0, (size_t)w * h * 4); }
static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 1}, .getDrawableInfo = get_drawable_info, .putImage = put_image, .getImage = get_image};
static const __DRIextension *loader_exts[] = {&swrast_loader.base, NULL};

static void *(*get_proc)(const char *);
#define GLF(ret, name, ...) ret (*p_##name)(__VA_ARGS__) = (ret (*)(__VA_ARGS__))get_proc(#name); if (!p_##name) { fprintf(stderr, "no %s\n", #name); return 3; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: mesa_raster in.bin out.bin\n"); return 1; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    int32_t hdr[5];
    float mvp[16];
    if (fread(hdr, 4, 5, fi) != 5 || fread(mvp, 4, 16, fi) != 16) return 1;
    const int W = hdr[0], H = hdr[1], nv = hdr[2], vs_len = hdr[3], fs_len = hdr[4];
    float *verts = malloc(sizeof(float) * 3 * nv), *attrs = malloc(sizeof(float) * 3 * nv);
    if (fread(verts, 4, 3 * (size_t)nv, fi) != 3 * (size_t)nv || fread(attrs, 4, 3 * (size_t)nv, fi) != 3 * (size_t)nv) return 1;
    char *vs_src = calloc(vs_len + 1, 1), *fs_src = calloc(fs_len + 1, 1);
    if (fread(vs_src, 1, vs_len, fi) != (size_t)vs_len || fread(fs_src, 1, fs_len, fi) != (size_t)fs_len) return 1;
    fclose(fi);

    setenv("LIBGL_ALWAYS_SOFTWARE", "1", 0);
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    const char *drv = getenv("MESA_SWRAST_PATH") ? getenv("MESA_SWRAST_PATH") : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!glapi || !h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    get_proc = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    const __DRIextension **(*get_exts)(void) = (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get_proc || !get_exts) { fprintf(stderr, "missing entry points\n"); return 2; }
    const __DRIextension **exts = get_exts();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { fprintf(stderr, "no core / swrast v4 extension\n"); return 2; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!screen || !configs) { fprintf(stderr, "createNewScreen2 failed\n"); return 2; }
    const __DRIconfig *cfg = configs[0];
    for (int i = 0; configs[i]; ++i) {          /* any RGBA8 + depth config will do: rendering goes to an FBO */
        unsigned r = 0, d = 0, db = 0;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &d);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && d == 24 && !db) { cfg = configs[i]; break; }
    }
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed (%u)\n", err); return 2; }
    __DRIdrawable *draw = swrast->createNewDrawable(screen, cfg, NULL);
    if (!draw || !core->bindContext(ctx, draw, draw)) { fprintf(stderr, "bindContext failed\n"); return 2; }

    GLF(const GLubyte *, glGetString, GLenum)
    GLF(GLuint, glCreateShader, GLenum) GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *) GLF(void, glCompileShader, GLuint)
    GLF(void, glGetShaderiv, GLuint, GLenum, GLint *) GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *)
    GLF(GLuint, glCreateProgram, void) GLF(void, glAttachShader, GLuint, GLuint) GLF(void, glLinkProgram, GLuint) GLF(void, glGetProgramiv, GLuint, GLenum, GLint *)
    GLF(void, glUseProgram, GLuint) GLF(GLint, glGetUniformLocation, GLuint, const GLchar *) GLF(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *)
    GLF(void, glGenVertexArrays, GLsizei, GLuint *) GLF(void, glBindVertexArray, GLuint) GLF(void, glGenBuffers, GLsizei, GLuint *) GLF(void, glBindBuffer, GLenum, GLuint)
    GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum) GLF(void, glEnableVertexAttribArray, GLuint)
    GLF(void, glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *)
    GLF(void, glGenFramebuffers, GLsizei, GLuint *) GLF(void, glBindFramebuffer, GLenum, GLuint) GLF(void, glGenTextures, GLsizei, GLuint *) GLF(void, glBindTexture, GLenum, GLuint)
    GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *) GLF(void, glTexParameteri, GLenum, GLenum, GLint)
    GLF(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint) GLF(void, glGenRenderbuffers, GLsizei, GLuint *) GLF(void, glBindRenderbuffer, GLenum, GLuint)
    GLF(void, glRenderbufferStorage, GLenum, GLenum, GLsizei, GLsizei) GLF(void, glFramebufferRenderbuffer, GLenum, GLenum, GLenum, GLuint)
    GLF(GLenum, glCheckFramebufferStatus, GLenum) GLF(void, glViewport, GLint, GLint, GLsizei, GLsizei) GLF(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat)
    GLF(void, glClear, GLbitfield) GLF(void, glEnable, GLenum) GLF(void, glDrawArrays, GLenum, GLint, GLsizei)
    GLF(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *) GLF(GLenum, glGetError, void) GLF(void, glFinish, void)
    fprintf(stderr, "GL_RENDERER: %s | GL_VERSION: %s\n", p_glGetString(GL_RENDERER), p_glGetString(GL_VERSION));

    /* Renderer.__init__ (renderer.py:337-387) */
    GLuint vs = p_glCreateShader(GL_VERTEX_SHADER), fs = p_glCreateShader(GL_FRAGMENT_SHADER);
    const char *vsp = vs_src, *fsp = fs_src;
    p_glShaderSource(vs, 1, &vsp, NULL); p_glCompileShader(vs);
    p_glShaderSource(fs, 1, &fsp, NULL); p_glCompileShader(fs);
    GLint ok = 0;
    p_glGetShaderiv(vs, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[2048]; p_glGetShaderInfoLog(vs, 2048, NULL, log); fprintf(stderr, "vs: %s\n", log); return 4; }
    p_glGetShaderiv(fs, GL_COMPILE_STATUS, &ok);
    if (!ok) { char log[2048]; p_glGetShaderInfoLog(fs, 2048, NULL, log); fprintf(stderr, "fs: %s\n", log); return 4; }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, vs); p_glAttachShader(prog, fs); p_glLinkProgram(prog);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { fprintf(stderr, "link failed\n"); return 4; }
    p_glUseProgram(prog);
    p_glUniformMatrix4fv(p_glGetUniformLocation(prog, "mvp"), 1, GL_TRUE, mvp);        /* set_mvp_mat (:389-392) */
    GLuint vao, vbo[2], fbo, tex, rbo;
    p_glGenVertexArrays(1, &vao); p_glGenBuffers(2, vbo);
    p_glGenFramebuffers(1, &fbo); p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glGenTextures(1, &tex); p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, NULL);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR); p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    p_glGenRenderbuffers(1, &rbo); p_glBindRenderbuffer(GL_RENDERBUFFER, rbo);
    p_glRenderbufferStorage(GL_RENDERBUFFER, GL_DEPTH24_STENCIL8, W, H);
    p_glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_STENCIL_ATTACHMENT, GL_RENDERBUFFER, rbo);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "framebuffer incomplete\n"); return 4; }
    /* set_model (:400-428) */
    p_glBindVertexArray(vao);
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo[0]); p_glBufferData(GL_ARRAY_BUFFER, sizeof(float) * 3 * nv, verts, GL_STREAM_DRAW);
    p_glEnableVertexAttribArray(0); p_glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, NULL);
    p_glBindBuffer(GL_ARRAY_BUFFER, vbo[1]); p_glBufferData(GL_ARRAY_BUFFER, sizeof(float) * 3 * nv, attrs, GL_STREAM_DRAW);
    p_glEnableVertexAttribArray(1); p_glVertexAttribPointer(1, 3, GL_FLOAT, GL_FALSE, 0, NULL);
    /* render (:432-451); the reference's viewport is its (hidden) window's size = the image size */
    p_glViewport(0, 0, W, H);
    p_glClearColor(0, 0, 0, 0);
    p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    p_glEnable(GL_DEPTH_TEST);
    p_glEnable(GL_CULL_FACE);
    p_glDrawArrays(GL_TRIANGLES, 0, nv);
    float *out = malloc(sizeof(float) * 4 * (size_t)W * H);
    p_glReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, out);
    p_glFinish();
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { fprintf(stderr, "GL error 0x%x\n", e); return 5; }
    FILE *fo = fopen(argv[2], "wb");
    fwrite(out, sizeof(float), 4 * (size_t)W * H, fo);
    fclose(fo);
    return 0;
}
